// K34 shaped for LATENCY: small batches, shapes without a software pipeline, streams beyond the Infinity Cache,
// and the multi-step launch.
//
// The software pipeline of k_step_pipe.hip hides HBM latency by giving every wavefront several env groups to
// stream through a shallow register ring.  With few envs there is nothing to stream: at BASELINE configs[1]
// (4 096 envs x 8 x 36) it degenerates to 512 single-group wavefronts (2 per CU) that each walk 8 load units two at a
// time -- a chain of four dependent memory round trips, 7.7 us for 13 MB (rocprofv3, profiles/r02a_c2_kernel_stats.csv),
// where a bare read of the same bytes takes 2-3 us.  The kernels here are shaped for latency instead:
//   * a wavefront owns EPWT <= 4 envs (not 64/VP), so even 4 096 envs make 2 048 wavefronts (8 per CU);
//   * EVERY load of the wavefront -- its step() inputs, then all h_r / theta rows -- is issued before the
//     first use: one memory round trip per wavefront, then the reduce, step() and the stores;
//   * with one or two envs per wavefront the gain-independent third of step() (step_pre) runs while the rows are
//     in flight.
// Same arithmetic in the same order as the pipelined kernel (same tiling, same transposing butterfly, same
// step_pre / step_tail): results are bit-identical, which tests/test_entry_points_hip.py asserts.
//
// Round 3: the tiling is a FusedShape<V, G, NIT, MC> -- MC = 0 takes ANY even M with that (G, NIT) from the launch
// dimensions, so every V in {4, 8, 16} x even M <= 256 has a member of this family (the reference's RIS-element study
// runs M = 20 ... 120 at V = 8, plt/plt-ris.py:7, marl_train_bcd.py:421-423); and TK = "theta by index": the env's
// phase shifts are read as the BCD sweep's candidate indices (1 byte per element, state.theta_idx) and expanded
// through a 9-entry table in LDS instead of streaming the complex64 row the sweep would otherwise have to write.
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

// NT: h_r / theta loads carry the non-temporal hint (streams beyond the Infinity Cache: see launch_step_fused_lat)
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

// STAMP: diagnostic build only (RISVEC_DIAG, tools/lat_stamps.py): TJ.reward is a debug buffer that receives, per
// wavefront, the s_memrealtime (100 MHz) of: entry, all loads issued, cascade reduced (loads returned), step() done,
// stores drained.
template <class S, int EPWT, bool MULTI, bool STAMP = false, int POL = 0, bool TK = false>
__global__ void __launch_bounds__(kBlock)
k_step_fused_lat(Dims d, RisVecParams P, StepArgs A, int n_steps, RisVecTraj TJ) {
    constexpr bool NT = POL == 1, ALT = POL == 2;       // cache policy of the h_r / theta loads: default, non-temporal, default + alternating walk
    constexpr int V = S::V, VP = S::VP, G = S::G, NIT = S::NIT, VPP = S::VPP;
    constexpr int PC = S::PC, CHUNKS = S::CHUNKS, K = S::K;
    constexpr int NU = EPWT * CHUNKS;                          // load units of the wavefront's envs
    constexpr bool EARLY_PRE = !MULTI && EPWT <= 2;            // step_pre in the shadow of the memory round trip
    static_assert(EPWT >= 1 && EPWT <= S::EPW, "a wavefront holds at most 64/VP envs");
    __shared__ float s_img[kBlock / kWave][kWave * 2];
    __shared__ float2 s_ph[TK ? kBlock / kWave : 1][16];       // TK: the 8 candidate phasors + the integer 0, per wavefront

    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int gl = lane % G, gv = lane / G;
    int wid = __builtin_amdgcn_readfirstlane(blockIdx.x * (kBlock / kWave) + wave);
    if constexpr (ALT) {
        // (its own instantiation: as a run-time flag of every kernel of the family the two scalar instructions and the
        // argument they wait for cost the 8 192-env shard 8 %: 7.1 -> 7.8 us, bisected in round 3)
        // A stream a little larger than the Infinity Cache (1 ... 1.29 x) is read with the default cache policy and walked
        // in ALTERNATING directions from step to step (A.ping: the launcher's step parity): the lines the previous
        // launch touched last -- the ones still cached -- are the first this launch asks for, where a same-direction
        // walk of an LRU-like cache hits nothing.  Which wavefront serves which env changes nothing in the results.
        // Measured (tools/gpu_pingpong.sh, profiles/r03q_pingpong_band.txt, r03p_*; h_r + theta per step at E x 8 x 64, us
        // per step round-2 dispatch / this): 283 MB 49.9 / 48.6, 302 MB 59.1 / 53.7, 340 MB 70.2 / 62.0, 377 MB 70.1 / 70.6,
        // 528 MB 100.2 / 101.5, 1.2 GB 226 / 236.
        const int n_waves = (d.E + EPWT - 1) / EPWT;
        if (A.ping && wid < n_waves) wid = n_waves - 1 - wid;
    }
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

    // step() inputs first: they are used first (step_pre) or last (four envs per wavefront), and loads return in issue order
    StepIn in = load_step_in(d, A, e_mine, v_mine, active);

    const float4* __restrict__ h4 = reinterpret_cast<const float4*>(A.h_r);
    const float4* __restrict__ t4 = reinterpret_cast<const float4*>(A.theta);
    const float4* __restrict__ b4 = reinterpret_cast<const float4*>(A.b);
    const int e_last = d.E - 1;
    const int NP = S::np(d);                                   // complex pairs per row: a constant for compile-time shapes

    if constexpr (TK) {
        // candidate k of the BCD sweep: exp(j 2 pi k / 8) as the float32 values the sweep stores (ENV:169, 213); 8 = 0
        if (lane < 16) {
            const float r = 0.70710677f;
            const int k = lane & 7;
            float c = (k & 3) == 2 ? 0.f : ((k & 1) ? r : 1.f);
            float sn = (k & 3) == 0 ? 0.f : ((k & 1) ? r : 1.f);
            if (k >= 3 && k <= 5) c = -c;
            if (k >= 5) sn = -sn;
            s_ph[wave][lane] = lane < 8 ? make_float2(c, sn) : make_float2(0.f, 0.f);
        }
    }

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
        pcl[it] = (!S::RAGGED || p < NP) ? p : NP - 1;
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
        unsigned tkw[UB][NIT];                                 // TK: the two candidate indices of each theta pair
#pragma unroll
        for (int k = 0; k < UB; ++k) {
            const int ui = b0 + k;
            const int i = ui / CHUNKS, c = ui % CHUNKS;
            int e = e0 + i;
            e = e < e_last ? e : e_last;                       // tail: re-read the last env, masked later
            const float4* __restrict__ hb = h4 + (long long)e * (V * NP);
#pragma unroll
            for (int pc = 0; pc < PC; ++pc) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    u[k].h[pc][it] = lat_ld<NT>(hb + (row_off + (unsigned)pcl[it] + ((c * PC + pc) * VPP * NP)));
                }
            }
            if (c == 0) {
                if constexpr (TK) {
                    const uint8_t* __restrict__ kb = A.theta_k + (long long)e * A.theta_k_stride;
#pragma unroll
                    for (int it = 0; it < NIT; ++it) tkw[k][it] = *reinterpret_cast<const uint16_t*>(kb + 2 * pcl[it]);
                } else {
                    const float4* __restrict__ tb = t4 + (long long)e * NP;
#pragma unroll
                    for (int it = 0; it < NIT; ++it) u[k].t[it] = lat_ld<NT>(tb + pcl[it]);
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
                    float2 t0, t1;
                    if constexpr (TK) {
                        __builtin_amdgcn_wave_barrier();       // the wavefront's own table writes -> its reads
                        t0 = s_ph[wave][tkw[k][it] & 15u];
                        t1 = s_ph[wave][(tkw[k][it] >> 8) & 15u];
                    } else {
                        t0 = make_float2(u[k].t[it].x, u[k].t[it].y);
                        t1 = make_float2(u[k].t[it].z, u[k].t[it].w);
                    }
                    w0[it] = cmul(t0, make_float2(bq[it].x, bq[it].y));
                    w1[it] = cmul(t1, make_float2(bq[it].z, bq[it].w));
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

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
static int env_int(const char* name, int dflt) {
    const char* s = std::getenv(name);
    return s ? std::atoi(s) : dflt;
}

// envs per wavefront: as many (<= 4) as keep >= 8 wavefronts per CU in flight (measured: EXPERIMENTS.md, rounds 1-2 section 6)
static int lat_epwt(int n_envs) {
    static const int forced = env_int("RISVEC_LAT_EPW", -1);   // A/B switch; 0 disables the latency-shaped single-step kernel
    if (forced >= 0) return forced;
    const long long want = 8LL * num_cus();
    int e = 4;
    while (e > 1 && (long long)n_envs / e < want) e >>= 1;
    return e;
}

static long long stream_bytes_of(const RisVecState& s) { return (long long)s.n_envs * (8LL * s.n_veh * s.n_ris + 8LL * s.n_ris); }

// RISVEC_LAT_PINGPONG: 0 never / 1 always (tests) / 2 (default) when the stream lies in (1, 1.29] x the Infinity Cache
static bool alternate_walk(const RisVecState& s) {
    static const int mode = env_int("RISVEC_LAT_PINGPONG", 2);
    if (mode != 2) return mode == 1;
    const long long b = stream_bytes_of(s);
    return b > tuning().ic_bytes && b <= tuning().lat_nt_from;
}

// POL: 0 default cache policy, 1 non-temporal loads, 2 default policy + alternating walk (single-step kernels only)
template <class S, int EPWT, bool MULTI, int POL, bool TK>
static hipError_t launch_one(const RisVecState& s, const RisVecParams& p, const StepArgs& a0, int n_steps,
                             const RisVecTraj& tj, hipStream_t st) {
    static_assert(POL != 2 || !MULTI, "the alternating walk is a single-step form");
    const long long waves = ((long long)s.n_envs + EPWT - 1) / EPWT;
    const unsigned grid = (unsigned)((waves + kBlock / kWave - 1) / (kBlock / kWave));
    StepArgs a = a0;
    a.ping = POL == 2 ? (int)(a.counter & 1u) : 0;
    hipLaunchKernelGGL((k_step_fused_lat<S, EPWT, MULTI, false, POL, TK>), dim3(grid), dim3(kBlock), 0, st, dims_of(s), p, a,
                       n_steps, tj);
    const char* pol = POL == 1 ? ",NT" : (POL == 2 ? ",ALT" : "");
    if (S::FIXED)
        note_kernel("k_step_fused_lat<%d,%d,%d%s%s%s>", S::V, S::MC, EPWT, MULTI ? ",MULTI" : "", pol, TK ? ",TK" : "");
    else
        note_kernel("k_step_fused_lat<%d,M=%d(G=%d,NIT=%d),%d%s%s%s>", S::V, s.n_ris, S::G, S::NIT, EPWT, MULTI ? ",MULTI" : "",
                    pol, TK ? ",TK" : "");
    return hipGetLastError();
}

// One shape class, single step: EPWT in [EMIN, EMAX] by batch size (default cache policy), EMAX with the non-temporal
// hint beyond the Infinity Cache; HAS_TK: the theta-by-index forms of the EMAX kernel exist.
template <class S, int EMIN, int EMAX, bool HAS_TK>
static hipError_t launch_single(const RisVecState& s, const RisVecParams& p, const StepArgs& a, int epwt, bool nt,
                                hipStream_t st) {
    const RisVecTraj none{};
    const bool alt = !nt && alternate_walk(s);                 // (1, 1.29] x the Infinity Cache: the EMAX kernel, walked both ways
    if (a.theta_k) {
        if constexpr (HAS_TK) {
            if (nt) return launch_one<S, EMAX, false, 1, true>(s, p, a, 1, none, st);
            if (alt) return launch_one<S, EMAX, false, 2, true>(s, p, a, 1, none, st);
            return launch_one<S, EMAX, false, 0, true>(s, p, a, 1, none, st);
        } else {
            return hipErrorNotSupported;
        }
    }
    if (nt) return launch_one<S, EMAX, false, 1, false>(s, p, a, 1, none, st);
    if (alt) return launch_one<S, EMAX, false, 2, false>(s, p, a, 1, none, st);
    if (epwt > EMAX) epwt = EMAX;
    if (epwt < EMIN) epwt = EMIN;
    if constexpr (EMIN <= 1 && EMAX >= 1) { if (epwt == 1) return launch_one<S, 1, false, 0, false>(s, p, a, 1, none, st); }
    if constexpr (EMIN <= 2 && EMAX >= 2) { if (epwt == 2) return launch_one<S, 2, false, 0, false>(s, p, a, 1, none, st); }
    if constexpr (EMIN <= 4 && EMAX >= 4) { if (epwt >= 3) return launch_one<S, 4, false, 0, false>(s, p, a, 1, none, st); }
    return hipErrorNotSupported;
}

template <class S>
static hipError_t launch_multi_shape(const RisVecState& s, const RisVecParams& p, const StepArgs& a, int n_steps,
                                     const RisVecTraj& tj, int epwt, hipStream_t st) {
    switch (epwt) {
        case 1: return launch_one<S, 1, true, 0, false>(s, p, a, n_steps, tj, st);
        case 2: return launch_one<S, 2, true, 0, false>(s, p, a, n_steps, tj, st);
        case 4: return launch_one<S, 4, true, 0, false>(s, p, a, n_steps, tj, st);
        case 8:
            // all 64 lanes stepping: only worth its registers where the step loop dominates (the T-step launch)
            if constexpr (S::EPW >= 8) return launch_one<S, 8, true, 0, false>(s, p, a, n_steps, tj, st);
            return hipErrorNotSupported;
        default: return hipErrorNotSupported;
    }
}

// compile-time shapes (the BASELINE configurations and the reference driver's default M = 40)
using S8x64 = FusedShape<8, 32, 1, 64>;
using S8x36 = FusedShape<8, 32, 1, 36>;
using S8x40 = FusedShape<8, 32, 1, 40>;
using S4x16 = FusedShape<4, 16, 1, 16>;
using S16x64 = FusedShape<16, 32, 1, 64>;
using S16x256 = FusedShape<16, 64, 2, 256>;

// Shapes with a member of this family: V in {4, 8, 16}, even M <= 256.
bool step_fused_lat_covers(int V, int M) {
    return (V == 4 || V == 8 || V == 16) && (M & 1) == 0 && M >= 2 && M <= 256;
}

// Shapes whose fused step can read theta as the BCD sweep's candidate indices (RISVEC_STEP_THETA_BY_INDEX): every
// shape this family covers (the EMAX-envs-per-wavefront member of each shape has a TK form)
bool theta_by_index_supported(int V, int M) { return step_fused_lat_covers(V, M); }

// Single step.  Which kernel, in order:
//   1. h_r + theta of one step beyond lat_nt_from (1.29 x the Infinity Cache): EMAX envs per wavefront, non-temporal
//      loads -- many short hardware-dispatched wavefronts with every request up front are what the best pure reader
//      looks like, and they beat the pipeline's 2 048 long-lived wavefronts by 1-4 % there (tools/gpu_latnt.sh,
//      profiles/r02t_lat_nt_experiment.txt, us per step pipeline / this: 65 536 envs 59.3 / 58.5, 131 072 115.8 / 111.9,
//      262 144 223.6 / 217.1; with the default cache policy it loses 7-12 % at those sizes).
//   2. shapes with a software pipeline (8 x 64, 4 x 16: up to 24 wavefronts per CU of 4 envs = 24 576 envs on 256 CUs;
//      16 x 64: 8 per CU = 8 192 envs): this kernel while the batch is that small (tools/gpu_crossover.sh,
//      profiles/r02t_lat_vs_pipe.txt, us per step this / pipeline at 8 x 64: 8 192 envs 7.3 / 9.4, 12 288 11.5 / 12.5,
//      20 480 17.9 / 18.7, 24 576 21.2 / 21.1, 32 768 27.5 / 26.9), the pipeline above.
//   3. everything else (M = 36 / 40: profiles/r02u_lat_vs_pipe_ragged.txt; 16 x 256; the run-time-M members): here at
//      every size.
// RISVEC_LAT_MAX_ENVS (0: never this kernel where a pipeline exists), RISVEC_LAT_EPW, RISVEC_LAT_NT (0 never / 1 always /
// 2 by size) and RISVEC_LAT_NT_MB override for A/Bs and the bit-identity tests.
hipError_t launch_step_fused_lat(const RisVecState& s, const RisVecParams& p, const StepArgs& a, hipStream_t st) {
    const int V = s.n_veh, M = s.n_ris;
    if (!step_fused_lat_covers(V, M)) return hipErrorNotSupported;
    static const long long forced_limit = [] { const char* e = std::getenv("RISVEC_LAT_MAX_ENVS"); return e ? std::atoll(e) : -1LL; }();
    static const int nt_mode = env_int("RISVEC_LAT_NT", 2);
    const long long stream_bytes = stream_bytes_of(s);
    const bool off = forced_limit == 0;                        // the family is switched off where a pipeline exists
    const bool nt = !off && (nt_mode == 1 || (nt_mode == 2 && stream_bytes > tuning().lat_nt_from));
    // between the cache size and lat_nt_from: this kernel with the default policy and the alternating walk (launch_one)
    const bool band = !off && !nt && nt_mode != 1 && alternate_walk(s) && stream_bytes > tuning().ic_bytes;
    const int epwt = lat_epwt(s.n_envs);
    if (epwt <= 0) return hipErrorNotSupported;
    // wavefronts per CU (of 4 envs) up to which this kernel beats the shape's software pipeline
    auto below = [&](int waves_per_cu) {
        const long long limit = forced_limit >= 0 ? forced_limit : (long long)waves_per_cu * num_cus() * 4;
        return (long long)s.n_envs <= limit;
    };
#ifdef RISVEC_DIAG
    // diagnostic library only (make diag -> librisvec_diag.so, tools/lat_stamps.py): the s_memrealtime build of the kernel
    if (const char* dbg = std::getenv("RISVEC_LAT_STAMPS_PTR")) {
        if (V == 8 && M == 36 && epwt == 2) {
            const RisVecTraj tj{reinterpret_cast<float*>(std::strtoull(dbg, nullptr, 0)), nullptr, nullptr};
            const long long waves = ((long long)s.n_envs + 1) / 2;
            hipLaunchKernelGGL((k_step_fused_lat<S8x36, 2, false, true>), dim3((unsigned)((waves + 3) / 4)), dim3(kBlock), 0, st,
                               dims_of(s), p, a, 1, tj);
            return hipGetLastError();
        }
    }
#endif
    if (V == 8 && M == 64) {
        if (!nt && !band && !a.theta_k && !below(24)) return hipErrorNotSupported;
        return launch_single<S8x64, 1, 4, true>(s, p, a, epwt, nt, st);
    }
    if (V == 4 && M == 16) {
        if (!nt && !band && !a.theta_k && !below(24)) return hipErrorNotSupported;
        return launch_single<S4x16, 1, 4, true>(s, p, a, epwt, nt, st);
    }
    if (V == 16 && M == 64) {
        // (tools/gpu_v16.sh, us per step pipeline / this: 4 096 envs 8.0 / 7.0, 8 192 14.5 / 14.5, from 16 384 the pipeline wins)
        if (!below(8) && !a.theta_k) return hipErrorNotSupported;
        return launch_single<S16x64, 1, 4, true>(s, p, a, epwt, false, st);
    }
    if (off && ((V == 8 && (M == 36 || M == 40)) || (V == 16 && M == 256)))
        return hipErrorNotSupported;                           // bit-identity tests: force the pipeline where one exists
    if (V == 8 && M == 36) return launch_single<S8x36, 1, 4, true>(s, p, a, epwt, nt, st);
    if (V == 8 && M == 40) return launch_single<S8x40, 1, 4, true>(s, p, a, epwt, nt, st);
    // 16 x 256 (one env = 34 loads per lane): one env per wavefront at every size (tools/gpu_v16.sh: 2 048 envs
    // 14.6 -> 13.5 us, 4 096 25.3 -> 23.4; BASELINE configs[4] with the BCD sweep 251.5 -> 233 us per step)
    if (V == 16 && M == 256) return launch_single<S16x256, 1, 1, true>(s, p, a, 1, nt, st);
    // run-time M: the member with this (V, G, NIT)
    const int g = fused_g(V, M), nit = fused_nit(V, M);
#define RISVEC_RT(VV, GG, NN, EMIN, EMAX) \
    if (V == VV && g == GG && nit == NN) return launch_single<FusedShape<VV, GG, NN, 0>, EMIN, EMAX, true>(s, p, a, epwt, nt, st);
    RISVEC_RT(8, 8, 1, 2, 4) RISVEC_RT(8, 16, 1, 2, 4) RISVEC_RT(8, 32, 1, 2, 4) RISVEC_RT(8, 64, 1, 2, 4) RISVEC_RT(8, 64, 2, 2, 4)
    RISVEC_RT(4, 16, 1, 2, 4) RISVEC_RT(4, 32, 1, 2, 4) RISVEC_RT(4, 64, 1, 2, 4) RISVEC_RT(4, 64, 2, 2, 4)
    RISVEC_RT(16, 8, 1, 2, 4) RISVEC_RT(16, 16, 1, 2, 4) RISVEC_RT(16, 32, 1, 2, 4) RISVEC_RT(16, 64, 1, 1, 2) RISVEC_RT(16, 64, 2, 1, 1)
#undef RISVEC_RT
    return hipErrorNotSupported;
}

// n_steps consecutive fused steps in one launch; hipErrorNotSupported when the shape has no
// compile-time kernel (the caller then issues one fused launch + one launch of k_step_multi).
hipError_t launch_step_fused_multi(const RisVecState& s, const RisVecParams& p, const StepArgs& a, int n_steps,
                                   const RisVecTraj* traj, hipStream_t st) {
    // The step loop is instruction-issue-bound (one step() is ~400 dependent-ish vector instructions per
    // wavefront whatever the number of active lanes), so a wavefront should carry as many envs as it has
    // lanes for -- as long as every SIMD still gets a wavefront (4 per CU).
    int epw = kWave / pow2_ceil(s.n_veh);
    if (epw > 8) epw = 8;
    static const int forced = env_int("RISVEC_MULTI_EPW", 0);
    int epwt = epw;
    while (epwt > 1 && (long long)s.n_envs / epwt < 4LL * num_cus()) epwt >>= 1;
    if (forced > 0) epwt = forced;
    const RisVecTraj tj = traj ? *traj : RisVecTraj{nullptr, nullptr, nullptr};
    const int V = s.n_veh, M = s.n_ris;
    if (V == 8 && M == 64) return launch_multi_shape<S8x64>(s, p, a, n_steps, tj, epwt, st);
    if (V == 8 && M == 36) return launch_multi_shape<S8x36>(s, p, a, n_steps, tj, epwt, st);
    if (V == 8 && M == 40) return launch_multi_shape<S8x40>(s, p, a, n_steps, tj, epwt, st);
    if (V == 4 && M == 16) return launch_multi_shape<S4x16>(s, p, a, n_steps, tj, epwt, st);
    return hipErrorNotSupported;
}

}  // namespace risvec
