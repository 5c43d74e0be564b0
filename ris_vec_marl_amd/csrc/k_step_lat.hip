// K34 for SMALL batches, and the multi-step launch.
//
// The software pipeline of k_step_pipe.hip hides HBM latency by giving every wavefront several env
// groups to stream through a shallow register ring.  With few envs there is nothing to stream: at
// BASELINE configs[1] (4 096 envs x 8 x 36) it degenerates to 512 single-group wavefronts (2 per CU)
// that each walk 8 load units two at a time -- a chain of four dependent memory round trips, 7.7 us
// for 13 MB (rocprofv3, profiles/r02a_c2_kernel_stats.csv), where a bare read of the same bytes takes
// 2-3 us.  The kernels here are shaped for latency instead:
//   * a wavefront owns EPWT <= 4 envs (not 64/VP), so even 4 096 envs make 2 048 wavefronts (8 per CU);
//   * EVERY load of the wavefront -- its step() inputs, then all h_r / theta rows -- is issued before the
//     first use: one memory round trip per wavefront, then the reduce, step() and the stores.
// Same arithmetic in the same order as the pipelined kernel (same PipeShape, same transposing butterfly,
// same step_core): results are bit-identical, which tests/test_entry_points_hip.py asserts.
//
// k_step_fused_lat<.., MULTI = true> is the T-step launch of SURVEY 7 ("launch latency"): n_steps
// consecutive step() calls of the driver loop (marl_train_bcd.py:1304-1611 with the groups frozen, as
// they are inside an episode) in ONE launch.  Envs never interact, so a wavefront simply keeps its
// envs' queues (DataBuf, mec_queue_cycles) in registers and walks the steps: actions are read from
// actions[t] (prefetched one step ahead), the Philox counter advances by one per step, each step's
// trajectory record (reward / obs / metrics) goes to its slice of the trajectory buffers, and the
// env's own tensors receive the last step's outputs -- bit-identical to n_steps single launches.
// h_r and theta cannot change inside a launch, so the cascaded gains are computed once.
//
// Reference: Simulation-MARL-BCD/Environment.py update_channel_gains ENV:263-273 + step ENV:547-731.
#include "risvec_pipe.hpp"

namespace risvec {

// STAMP: diagnostic build (tools/lat_stamps.py): TJ.reward is a debug buffer that receives, per wavefront, the
// s_memrealtime (100 MHz) of: entry, all loads issued, cascade reduced (loads returned), step() done, stores drained.
// NT: h_r / theta loads carry the non-temporal hint (streams beyond the 256 MiB Infinity Cache: see launch_step_fused_lat)
template <bool NT>
__device__ __forceinline__ float4 lat_ld(const float4* p) {
    if constexpr (NT) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
        return make_float4(t.x, t.y, t.z, t.w);
    } else {
        return *p;
    }
}

template <int V, int M, int EPWT, bool MULTI, bool STAMP = false, bool NT = false>
__global__ void __launch_bounds__(kBlock)
k_step_fused_lat(Dims d, RisVecParams P, StepArgs A, int n_steps, RisVecTraj TJ) {
    using S = PipeShape<V, M>;
    constexpr int VP = S::VP, NP = S::NP, G = S::G, NIT = S::NIT, VPP = S::VPP;
    constexpr int PC = S::PC, CHUNKS = S::CHUNKS, K = S::K;
    constexpr int NU = EPWT * CHUNKS;                          // load units of the wavefront's envs
    constexpr bool EARLY_PRE = !MULTI && EPWT <= 2;            // step_pre in the shadow of the memory round trip
    static_assert(EPWT >= 1 && EPWT <= S::EPW, "a wavefront holds at most 64/VP envs");
    __shared__ float s_img[kBlock / kWave][kWave * 2];

    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int gl = lane % G, gv = lane / G;
    const int wid = __builtin_amdgcn_readfirstlane(blockIdx.x * (kBlock / kWave) + wave);
    const int e0 = wid * EPWT;
    unsigned long long ts[5] = {0, 0, 0, 0, 0};
    if constexpr (STAMP) ts[0] = __builtin_amdgcn_s_memrealtime();
    // No early exit for the (at most three) surplus wavefronts of the last workgroup: a branch on d.E here costs every
    // wavefront a scalar-load round trip (cold scalar cache, ~0.35 us) BEFORE the kernel-argument pointers are even
    // requested.  Surplus wavefronts run with every lane inactive; all their addresses are clamped to the last env.
    const int v_mine = lane % VP, e_mine = e0 + lane / VP;
    const bool active = (lane / VP) < EPWT && e_mine < d.E;
    // step()'s scalar parameters with the pointers, in the kernel's first scalar round trip -- where one or two envs per
    // wavefront make the kernel a pure latency chain (same box, interleaved: configs[1] 5.09 -> 4.83 us).  With four
    // envs per wavefront the same lines cost more at the front than they save inside step() (the configs[3] shard:
    // 7.27 -> 7.68 us), and the T-step launch pays for them once per T steps either way.
    if constexpr (!MULTI && EPWT <= 2) RISVEC_ARGS_IN_ONE_TRIP(RISVEC_STEP_PARAMS(P));

    // step() inputs first: they are used last, and loads return in issue order
    StepIn in = load_step_in(d, A, e_mine, v_mine, active);

    const float4* __restrict__ h4 = reinterpret_cast<const float4*>(A.h_r);
    const float4* __restrict__ t4 = reinterpret_cast<const float4*>(A.theta);
    const float4* __restrict__ b4 = reinterpret_cast<const float4*>(A.b);
    const int e_last = d.E - 1;

    float4 bq[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int p = gl + it * G;
        const float4 x = b4[p < NP ? p : NP - 1];
        bq[it] = p < NP ? x : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // ragged rows: lanes past the end re-read the row's last float4 (see k_step_pipe.hip) -- a skipped load made the
    // compiler put `s_waitcnt vmcnt(0)` between the first env's loads and the second's: two round trips, not one
    int pcl[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int p = gl + it * G;
        pcl[it] = (NP % G == 0 || p < NP) ? p : NP - 1;
    }
    const unsigned row_off = (unsigned)(gv * NP);
    // Units are loaded in batches of up to 4 (all of them when the wavefront owns <= 4 envs: one memory
    // round trip); the multi-step form may own 64/VP envs, where the gain phase is amortised over the steps.
    if constexpr (STAMP) ts[1] = __builtin_amdgcn_s_memrealtime();
    constexpr int UB = NU > 4 ? 4 : NU;
    static_assert(NU % UB == 0, "unit batches must tile the wavefront's units");
    float2 w0[NIT], w1[NIT];
    StepPre pre{};
#pragma unroll
    for (int b0 = 0; b0 < NU; b0 += UB) {
        Unit<PC, NIT> u[UB];
#pragma unroll
        for (int k = 0; k < UB; ++k) {
            const int ui = b0 + k;
            const int i = ui / CHUNKS, c = ui % CHUNKS;
            int e = e0 + i;
            e = e < e_last ? e : e_last;                       // tail: re-read the last env, masked later
            const float4* __restrict__ hb = h4 + (long long)e * (V * NP);
            const float4* __restrict__ tb = t4 + (long long)e * NP;
#pragma unroll
            for (int pc = 0; pc < PC; ++pc) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    u[k].h[pc][it] = lat_ld<NT>(hb + (row_off + (unsigned)pcl[it] + ((c * PC + pc) * VPP * NP)));
                }
            }
            if (c == 0) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    u[k].t[it] = lat_ld<NT>(tb + pcl[it]);
                }
            }
        }
        // The gain-independent third of step() (action map, projection, local CPU terms, the Philox / Poisson arrival
        // draw) runs HERE, behind the last request and in front of the first wait on h_r: its inputs were requested
        // first and arrive first, and the SIMD has nothing else to issue until the rows are back.
        // (one or two envs per wavefront only: with four, rows of the first env are back before the last request has
        // left, the reduce has to start consuming them at once, and the same instructions in front of it measured
        // +7 % at the BASELINE configs[3] shard)
        if constexpr (EARLY_PRE) {
            if (b0 == 0) pre = step_pre(d, P, A, e_mine, v_mine, active, in);
        }
        // 16 x 256 (one env = 34 loads per lane): left alone the compiler starts reducing before the last request is
        // out, to stay at 123 registers; with every request up front (168 registers, 3 wavefronts per SIMD) BASELINE
        // configs[4] is 0.8 % faster.  At 8 x 64 the same barrier measured neutral, at the configs[3] shard slower.
        if constexpr (NT && V == 16) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < UB; ++k) {
            const int ui = b0 + k;
            const int i = ui / CHUNKS, c = ui % CHUNKS;
            if (c == 0) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    w0[it] = cmul(make_float2(u[k].t[it].x, u[k].t[it].y), make_float2(bq[it].x, bq[it].y));
                    w1[it] = cmul(make_float2(u[k].t[it].z, u[k].t[it].w), make_float2(bq[it].z, bq[it].w));
                }
            }
            float val[8];
#pragma unroll
            for (int pc = 0; pc < PC; ++pc) {
                float2 acc = make_float2(0.f, 0.f);
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    acc = cfma(make_float2(u[k].h[pc][it].x, u[k].h[pc][it].y), w0[it], acc);
                    acc = cfma(make_float2(u[k].h[pc][it].z, u[k].h[pc][it].w), w1[it], acc);
                }
                val[2 * pc] = acc.x;
                val[2 * pc + 1] = acc.y;
            }
            treduce<K, G / 2>(val, gl);
            if (gl % S::WSTRIDE == 0) {
                const int j = gl / S::WSTRIDE;
                const int v = (c * PC + (j >> 1)) * VPP + gv;
                s_img[wave][(i * VP + v) * 2 + (j & 1)] = val[0];
            }
        }
    }
    __builtin_amdgcn_wave_barrier();                           // own LDS writes -> own reads (in order per wave)
    if constexpr (STAMP) ts[2] = __builtin_amdgcn_s_memrealtime();
    const float2 img = *reinterpret_cast<const float2*>(&s_img[wave][lane * 2]);
    const long long idx = (long long)e_mine * V + v_mine;
    float g = 0.f;
    if (active) {
        g = gain_from_img(img, in.pl, A.h_d, idx);
        A.gain[idx] = g;
    }
    if constexpr (!EARLY_PRE && !MULTI) pre = step_pre(d, P, A, e_mine, v_mine, active, in);
    if constexpr (STAMP) {
        step_tail<VP, false, true>(d, P, A, e_mine, v_mine, active, g, in, pre);
        asm volatile("" ::: "memory");
        ts[3] = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ts[4] = __builtin_amdgcn_s_memrealtime();
        if (lane == 0 && e0 < d.E) {
            unsigned long long* dbg = reinterpret_cast<unsigned long long*>(TJ.reward) + (long long)wid * 5;
            dbg[0] = ts[0]; dbg[1] = ts[1]; dbg[2] = ts[2]; dbg[3] = ts[3]; dbg[4] = ts[4];
        }
    } else if constexpr (!MULTI) {
        step_tail<VP, false, true>(d, P, A, e_mine, v_mine, active, g, in, pre);
    } else {
        multi_step_loop<VP>(d, P, A, TJ, e_mine, v_mine, active, g, in, n_steps);
    }
}

// envs per wavefront: as many as keep >= 8 wavefronts per CU in flight (measured: see DESIGN.md section 6)
static int lat_epwt(int n_envs, int max_epw) {
    static const int forced = [] { const char* s = std::getenv("RISVEC_LAT_EPW"); return s ? std::atoi(s) : -1; }();
    if (forced >= 0) return forced;                            // 0 disables the latency-shaped single-step kernel
    const long long want = 8LL * num_cus();
    int e = 4;
    while (e > 1 && (long long)n_envs / e < want) e >>= 1;
    return e < max_epw ? e : max_epw;
}

template <int V, int M, int EPWT, bool MULTI>
static hipError_t launch_lat_shape(const RisVecState& s, const RisVecParams& p, const StepArgs& a, int n_steps,
                                   const RisVecTraj& tj, hipStream_t st) {
    const long long waves = ((long long)s.n_envs + EPWT - 1) / EPWT;
    const unsigned grid = (unsigned)((waves + kBlock / kWave - 1) / (kBlock / kWave));
    hipLaunchKernelGGL((k_step_fused_lat<V, M, EPWT, MULTI>), dim3(grid), dim3(kBlock), 0, st, dims_of(s), p, a, n_steps, tj);
    return hipGetLastError();
}

template <int V, int M, bool MULTI>
static hipError_t launch_lat_vm(const RisVecState& s, const RisVecParams& p, const StepArgs& a, int n_steps,
                                const RisVecTraj& tj, int epwt, hipStream_t st) {
    switch (epwt) {
        case 1: return launch_lat_shape<V, M, 1, MULTI>(s, p, a, n_steps, tj, st);
        case 2: return launch_lat_shape<V, M, 2, MULTI>(s, p, a, n_steps, tj, st);
        case 4: return launch_lat_shape<V, M, 4, MULTI>(s, p, a, n_steps, tj, st);
        case 8:
            // all 64 lanes stepping: only worth its registers where the step loop dominates (the T-step launch)
            if constexpr (MULTI) return launch_lat_shape<V, M, 8, MULTI>(s, p, a, n_steps, tj, st);
            return hipErrorNotSupported;
        default: return hipErrorNotSupported;
    }
}

// EPWT envs per wavefront (4; 1 at 16 x 256, where one env is already 34 KB in flight), single step, non-temporal loads
template <int V, int M, int EPWT = 4>
static hipError_t launch_lat_nt(const RisVecState& s, const RisVecParams& p, const StepArgs& a, hipStream_t st) {
    const long long waves = ((long long)s.n_envs + EPWT - 1) / EPWT;
    const unsigned grid = (unsigned)((waves + kBlock / kWave - 1) / (kBlock / kWave));
    const RisVecTraj none{nullptr, nullptr, nullptr};
    hipLaunchKernelGGL((k_step_fused_lat<V, M, EPWT, false, false, true>), dim3(grid), dim3(kBlock), 0, st, dims_of(s), p, a, 1, none);
    return hipGetLastError();
}

static hipError_t dispatch_lat_nt(const RisVecState& s, const RisVecParams& p, const StepArgs& a, hipStream_t st) {
    const int V = s.n_veh, M = s.n_ris;
    if (V == 8 && M == 64) return launch_lat_nt<8, 64>(s, p, a, st);
    if (V == 8 && M == 36) return launch_lat_nt<8, 36>(s, p, a, st);
    if (V == 8 && M == 40) return launch_lat_nt<8, 40>(s, p, a, st);
    if (V == 4 && M == 16) return launch_lat_nt<4, 16>(s, p, a, st);
    if (V == 16 && M == 256) return launch_lat_nt<16, 256, 1>(s, p, a, st);
    return hipErrorNotSupported;
}

template <bool MULTI>
static hipError_t dispatch_lat(const RisVecState& s, const RisVecParams& p, const StepArgs& a, int n_steps,
                               const RisVecTraj& tj, int epwt, hipStream_t st) {
    const int V = s.n_veh, M = s.n_ris;
    if constexpr (!MULTI) {
        // 16 vehicles, single step (tools/gpu_v16.sh, us per step pipeline / this): 16 x 256 with one env per wavefront
        // 2 048 envs 14.6 / 13.5, 4 096 25.3 / 23.4, 7 168 40.1 / 38.7 (and the non-temporal form beyond 330 MB per step:
        // launch_step_fused_lat); 16 x 64: 4 096 envs 8.0 / 7.0, 8 192 14.5 / 14.5, from 16 384 the pipeline wins.
        if (V == 16 && M == 256) return launch_lat_shape<16, 256, 1, false>(s, p, a, 1, tj, st);
        if (V == 16 && M == 64 && s.n_envs <= 8192) {
            if (epwt >= 4) return launch_lat_shape<16, 64, 4, false>(s, p, a, 1, tj, st);
            if (epwt == 2) return launch_lat_shape<16, 64, 2, false>(s, p, a, 1, tj, st);
            return launch_lat_shape<16, 64, 1, false>(s, p, a, 1, tj, st);
        }
    }
    if (V == 8 && M == 64) return launch_lat_vm<8, 64, MULTI>(s, p, a, n_steps, tj, epwt, st);
    if (V == 8 && M == 36) return launch_lat_vm<8, 36, MULTI>(s, p, a, n_steps, tj, epwt, st);
    if (V == 8 && M == 40) return launch_lat_vm<8, 40, MULTI>(s, p, a, n_steps, tj, epwt, st);
    if (V == 4 && M == 16) return launch_lat_vm<4, 16, MULTI>(s, p, a, n_steps, tj, epwt, st);
    return hipErrorNotSupported;
}

// Single step, small and medium batches: taken instead of the software pipeline up to 24 576 envs.  Re-measured after
// the ragged-row fix (tools/gpu_crossover.sh, profiles/r02t_lat_vs_pipe.txt; us per step, this kernel / pipeline):
// 8 192 x 8 x 64  7.3 / 9.4, 12 288  11.5 / 12.5, 16 384  15.7 / 15.2, 20 480  17.9 / 18.7, 24 576  21.2 / 21.1,
// 32 768  27.5 / 26.9;  M = 36: 8 192  5.7 / 6.5, 18 432  11.9 / 13.8, 32 768  18.0 / 18.4.  Four envs per wavefront with
// every request up front keep up with the pipeline for as long as the batch is a few wavefronts per SIMD, and the
// hardware dispatcher balances 4-env wavefronts better than the pipeline's fixed 2 048 wavefronts balance env groups
// (18 432 envs = 2 304 groups = 1 152 wavefronts of two).
hipError_t launch_step_fused_lat(const RisVecState& s, const RisVecParams& p, const StepArgs& a, hipStream_t st) {
    const int vp = pow2_ceil(s.n_veh);
    const int epw = kWave / vp;
    static const long long limit = [] {                        // envs below which the latency shape wins
        const char* e = std::getenv("RISVEC_LAT_MAX_ENVS");
        return e ? std::atoll(e) : 24576LL;
    }();
    // Beyond the Infinity Cache (h_r + theta of one step > RISVEC_LAT_NT_MB = 330 MB; the pipeline switches to
    // non-temporal loads from 270 MB) this kernel comes back with the non-temporal hint: many short hardware-dispatched
    // wavefronts with every request up front are what the best pure reader looks like, and they beat the pipeline's
    // 2 048 long-lived wavefronts by 1-4 % there (tools/gpu_latnt.sh, profiles/r02t_lat_nt_experiment.txt, us per step
    // pipeline / this: 65 536 envs 59.3 / 58.5, 131 072 115.8 / 111.9, 262 144 223.6 / 217.1; with the default cache
    // policy it loses 7-12 % at those sizes, and inside the cache (24 576 < envs <= ~57 000) the pipeline stays).
    // BASELINE configs[4] (32 768 x 16 x 256, 1.1 GB per step, one env per wavefront): 251.5 -> 233 us per step with
    // the BCD sweep, i.e. the fused kernel 198 -> 180 us.
    // RISVEC_LAT_NT = 0 never / 1 always (tests) / 2 by size.
    static const int nt_mode = [] { const char* e = std::getenv("RISVEC_LAT_NT"); return e ? std::atoi(e) : 2; }();
    // (its own threshold: tools/gpu_nt_threshold.sh at 16 x 256, us per step default / non-temporal: 221 MB 33.4 / 35.8,
    // 294 MB 43.9 / 47.4, 368 MB 66.2 / 59.9, 441 MB 78.5 / 71.0, 588 MB 107.4 / 96.6 -- the crossover is near 330 MB)
    static const long long nt_from = [] {
        const char* e = std::getenv("RISVEC_LAT_NT_MB");
        return (e ? std::atoll(e) : 330LL) << 20;
    }();
    const long long stream_bytes = (long long)s.n_envs * (8LL * s.n_veh * s.n_ris + 8LL * s.n_ris);
    if (limit > 0 && epw >= 4 && (nt_mode == 1 || (nt_mode == 2 && stream_bytes > nt_from))) {
        const hipError_t err = dispatch_lat_nt(s, p, a, st);
        if (err != hipErrorNotSupported) return err;
    }
    // M = 36 / 40 (18 / 20 float4 per row on 32 lanes): this kernel at EVERY size -- the pipeline's unit loader wastes the
    // same lanes and gains nothing back (tools/gpu_crossover.sh, profiles/r02u_lat_vs_pipe_ragged.txt, us per step this
    // kernel / pipeline at M = 40: 32 768 envs 19.3 / 19.6, 65 536 34.2 / 34.7, 98 304 62.5 / 65.7; M = 36: 65 536 31.6 / 32.7)
    const bool ragged = s.n_veh == 8 && (s.n_ris == 36 || s.n_ris == 40) && limit > 0;
    if ((long long)s.n_envs > limit && !ragged) return hipErrorNotSupported;
    const int epwt = lat_epwt(s.n_envs, epw);
    if (epwt <= 0) return hipErrorNotSupported;
    const RisVecTraj none{nullptr, nullptr, nullptr};
#ifdef RISVEC_DIAG
    // diagnostic library only (make diag -> librisvec_diag.so, tools/lat_stamps.py): the s_memrealtime build of the kernel
    if (const char* dbg = std::getenv("RISVEC_LAT_STAMPS_PTR")) {
        if (s.n_veh == 8 && s.n_ris == 36 && epwt == 2) {
            const RisVecTraj tj{reinterpret_cast<float*>(std::strtoull(dbg, nullptr, 0)), nullptr, nullptr};
            const long long waves = ((long long)s.n_envs + 1) / 2;
            hipLaunchKernelGGL((k_step_fused_lat<8, 36, 2, false, true>), dim3((unsigned)((waves + 3) / 4)), dim3(kBlock), 0, st,
                               dims_of(s), p, a, 1, tj);
            return hipGetLastError();
        }
    }
#endif
    return dispatch_lat<false>(s, p, a, 1, none, epwt, st);
}

// n_steps consecutive fused steps in one launch; hipErrorNotSupported when the shape has no
// compile-time kernel (the caller then issues n_steps single launches).
hipError_t launch_step_fused_multi(const RisVecState& s, const RisVecParams& p, const StepArgs& a, int n_steps,
                                   const RisVecTraj* traj, hipStream_t st) {
    // The step loop is instruction-issue-bound (one step() is ~400 dependent-ish vector instructions per
    // wavefront whatever the number of active lanes), so a wavefront should carry as many envs as it has
    // lanes for -- as long as every SIMD still gets a wavefront (4 per CU).
    int epw = kWave / pow2_ceil(s.n_veh);
    if (epw > 8) epw = 8;
    static const int forced = [] { const char* e = std::getenv("RISVEC_MULTI_EPW"); return e ? std::atoi(e) : 0; }();
    int epwt = epw;
    while (epwt > 1 && (long long)s.n_envs / epwt < 4LL * num_cus()) epwt >>= 1;
    if (forced > 0) epwt = forced;
    const RisVecTraj tj = traj ? *traj : RisVecTraj{nullptr, nullptr, nullptr};
    return dispatch_lat<true>(s, p, a, n_steps, tj, epwt, st);
}

}  // namespace risvec
