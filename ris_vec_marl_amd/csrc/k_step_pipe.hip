// K34, compile-time shapes: the fused "RIS cascaded gains + step()" kernel as a software
// pipeline.  Same results as the generic k_step_fused (k_step.hip), restructured so the
// HBM stream never waits for the step arithmetic:
//
//   * a wave owns SEVERAL groups of 64/VP envs (grid-strided), not one: while it runs
//     step() for group g, the h_r/theta loads of group g+1 are already in flight;
//   * loads are issued D "units" ahead (a unit = up to 4 vehicle rows of one env, i.e. up to
//     8 x 16-byte loads per lane) into a register ring that is indexed at compile time;
//   * the K = 2*PC per-lane partial sums of a unit (re/im of PC rows) are reduced with a
//     TRANSPOSING butterfly: at each of the first log2(K) steps a lane hands half of its
//     values to its partner and keeps the other half, so K values cost K-1 exchanges
//     instead of K*log2(G); exchanges at distance 1, 2, 4, 8 are DPP (plain VALU);
//   * step() inputs are fetched at the top of the group, long before they are needed.
//
// Reference: Simulation-MARL-BCD/Environment.py update_channel_gains ENV:263-273 + step
// ENV:547-731 (see risvec_step.hpp).
#include <cstdlib>

#include "risvec_pipe.hpp"
#include "risvec_sarl.hpp"

namespace risvec {

struct SarlIn {
    float p0, p1, B, pl;
};

struct SarlCore {
    static const char* name() { return "SarlCore"; }
    using Params = RisVecSarlParams;
    using Args = SarlArgs;
    using In = SarlIn;
    static __device__ __forceinline__ In load(const Dims& d, const Args& A, int e, int v, bool active) {
        In in{0.f, 0.f, 0.f, 0.f};
        if (active) {
            const long long idx = (long long)e * d.V + v;
            in.p0 = A.action_power[(long long)e * 2 * d.V + v];
            in.p1 = A.action_power[(long long)e * 2 * d.V + d.V + v];
            in.B = A.data_buf[idx];
            in.pl = A.pl[idx];
        }
        return in;
    }
    static __device__ __forceinline__ void hold(const In& in) {
        asm volatile("" ::"v"(in.p0), "v"(in.p1), "v"(in.B), "v"(in.pl));
    }
    template <int VP>
    static __device__ __forceinline__ void run(const Dims& d, const Params& P, const Args& A, int e, int v,
                                               bool active, float2 img, const In& in) {
        float g = 0.f;
        if (active) {
            const long long idx = (long long)e * d.V + v;
            g = gain_from_img(img, in.pl, nullptr, idx);
            A.gain[idx] = g;
        }
        sarl_core<VP>(d, P, A, e, v, active, g, in.p0, in.p1, in.B);
    }
};

struct GainArgs {
    const float* pl;
    const float* h_r;
    const float* theta;
    const float* b;
    const float* h_d;
    float* gain;
};

struct GainCore {
    static const char* name() { return "GainCore"; }
    using Params = int;
    using Args = GainArgs;
    struct In { float pl; };
    static __device__ __forceinline__ In load(const Dims& d, const Args& A, int e, int v, bool active) {
        return In{active ? A.pl[(long long)e * d.V + v] : 0.f};
    }
    static __device__ __forceinline__ void hold(const In& in) { asm volatile("" ::"v"(in.pl)); }
    template <int VP>
    static __device__ __forceinline__ void run(const Dims& d, const Params&, const Args& A, int e, int v,
                                               bool active, float2 img, const In& in) {
        if (active) {
            const long long idx = (long long)e * d.V + v;
            A.gain[idx] = gain_from_img(img, in.pl, A.h_d, idx);
        }
    }
};

// float4 load with the non-temporal hint (global_load ... nt): for a stream far larger than the 256 MiB Infinity
// Cache, allocating every line on its way through only evicts what somebody else could have reused
__device__ __forceinline__ float4 ld_nt(const float4* p) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
    return make_float4(t.x, t.y, t.z, t.w);
}

// per_wave > 0: a wavefront owns the CONTIGUOUS groups [wid * per_wave, (wid + 1) * per_wave) -- it walks one
// contiguous piece of h_r from start to end; per_wave = 0: groups wid, wid + nw, ... (the round-1 form).
// NT: h_r / theta are read with the non-temporal hint.
template <int V, int M, int D, class Core, bool NT = false>
__global__ void __launch_bounds__(kBlock)
k_step_fused_pipe(Dims d, typename Core::Params P, typename Core::Args A, int n_groups_total, int per_wave) {
    using In = typename Core::In;
    using S = PipeShape<V, M>;
    constexpr int VP = S::VP, EPW = S::EPW, NP = S::NP, G = S::G, NIT = S::NIT, VPP = S::VPP;
    constexpr int PC = S::PC, CHUNKS = S::CHUNKS, UPG = S::UPG, K = S::K;
    static_assert(D <= UPG && UPG % D == 0, "ring depth must divide the units of a group");
    __shared__ float s_img[kBlock / kWave][kWave * 2];

    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int gl = lane % G, gv = lane / G;
    // wave-uniform by construction; tell the compiler so group indices live in SGPRs
    const int wid = __builtin_amdgcn_readfirstlane(blockIdx.x * (kBlock / kWave) + wave);
    const int nw = gridDim.x * (kBlock / kWave);
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
    // Ragged rows (NP % G != 0, i.e. M = 36 / 40: 18 / 20 float4 on 32 lanes): the lanes past the end of a row RE-READ
    // its last float4 instead of skipping the load.  A skipped load is a branch around the request with a zero fill
    // behind it, and the compiler can only order that fill against the loads in flight with `s_waitcnt vmcnt(0)` --
    // every unit then drains the whole ring (16 full drains per group in the M = 36 kernel, no counted wait left).
    // The re-read element shares its cache line with the row's last lane (no extra HBM traffic), and what those
    // lanes add to the sums is an exact +0: their b[m] is zero, so w0 / w1 are (signed) zeros, and +0 + (h * -0) = +0.
    int pcl[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int p = gl + it * G;
        pcl[it] = (NP % G == 0 || p < NP) ? p : NP - 1;
    }
    const unsigned row_off = (unsigned)(gv * NP);

    using U = Unit<PC, NIT>;
    U ring[D];

    auto load_unit = [&](U& u, int grp, int ui) {
        const int i = ui / CHUNKS, c = ui % CHUNKS;
        int e = grp * EPW + i;
        e = e < e_last ? e : e_last;                       // tail: re-read the last env, masked later
        // wave-uniform 64-bit base (SGPRs) + 32-bit lane offset + compile-time constant:
        // the address arithmetic stays off the vector ALU
        const float4* __restrict__ hb = h4 + (long long)e * (V * NP);
        const float4* __restrict__ tb = t4 + (long long)e * NP;
#pragma unroll
        for (int pc = 0; pc < PC; ++pc) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const float4* __restrict__ src = hb + (row_off + (unsigned)pcl[it] + ((c * PC + pc) * VPP * NP));
                u.h[pc][it] = NT ? ld_nt(src) : *src;
            }
        }
        if (c == 0) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                u.t[it] = NT ? ld_nt(tb + pcl[it]) : tb[pcl[it]];
            }
        }
    };

    // per_wave < 0 (experiment): the four wavefronts of a workgroup interleave over ONE contiguous range of
    // 4 * |per_wave| groups
    constexpr int WPB = kBlock / kWave;
    const int stride = per_wave > 0 ? 1 : (per_wave < 0 ? WPB : nw);
    int grp = per_wave > 0 ? wid * per_wave : (per_wave < 0 ? (wid / WPB) * (-per_wave * WPB) + (wid % WPB) : wid);
    int grp_end = per_wave > 0 ? grp + per_wave : (per_wave < 0 ? (wid / WPB + 1) * (-per_wave * WPB) : n_groups_total);
    grp_end = grp_end < n_groups_total ? grp_end : n_groups_total;
    // Fetch the kernel-argument pointers in the SAME scalar-load round trip as the exit condition: left alone, the
    // compiler loads n_groups_total first, waits, branches, and only then requests the pointers -- a second cold
    // scalar-cache miss (~0.35 us) in front of every wavefront's first HBM request.
    RISVEC_ARGS_IN_ONE_TRIP("s"(A.h_r), "s"(A.theta), "s"(A.b), "s"(n_groups_total), "s"(per_wave), "s"(d.E));
    if (grp >= grp_end) return;                            // whole wave: no cross-lane op is skipped
#pragma unroll
    for (int ui = 0; ui < D; ++ui) load_unit(ring[ui], grp, ui);
    const int v_mine = lane % VP;

    // One group: consume its UPG units (refilling the ring D units ahead, across the group
    // boundary), prefetch the NEXT group's per-lane inputs into `in_nx`, then run the core with
    // `in`.  Called alternately with (inA, inB) / (inB, inA) so no register copy - and hence
    // no wait on the prefetch - is needed at the loop boundary.
    auto do_group = [&](int g_cur, const In& in, In& in_nx) {
        // The prefetch is unconditional (straight-line code keeps the compiler's vmcnt
        // bookkeeping exact): a wave on its last group "prefetches" group 0 instead, which
        // every such wave shares, so those requests are served by L2 and cost no HBM traffic.
        const int nxt = (g_cur + stride < grp_end) ? g_cur + stride : 0;
        const int e_mine = g_cur * EPW + lane / VP;
        const bool active = e_mine < d.E;
        // Take the wait for this group's per-lane inputs HERE (they were requested a whole
        // group ago), not at their first use inside the core, where the compiler could only
        // express it as vmcnt(0) and would drain the next group's prefetch with it.
        Core::hold(in);

        float2 w0[NIT], w1[NIT];
#pragma unroll
        for (int ui = 0; ui < UPG; ++ui) {
            U& u = ring[ui % D];
            const int i = ui / CHUNKS, c = ui % CHUNKS;
            if (c == 0) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    w0[it] = cmul(make_float2(u.t[it].x, u.t[it].y), make_float2(bq[it].x, bq[it].y));
                    w1[it] = cmul(make_float2(u.t[it].z, u.t[it].w), make_float2(bq[it].z, bq[it].w));
                }
            }
            float val[8];
#pragma unroll
            for (int pc = 0; pc < PC; ++pc) {
                float2 acc = make_float2(0.f, 0.f);
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    acc = cfma(make_float2(u.h[pc][it].x, u.h[pc][it].y), w0[it], acc);
                    acc = cfma(make_float2(u.h[pc][it].z, u.h[pc][it].w), w1[it], acc);
                }
                val[2 * pc] = acc.x;
                val[2 * pc + 1] = acc.y;
            }
            treduce<K, G / 2>(val, gl);
            if (gl % S::WSTRIDE == 0) {
                const int j = gl / S::WSTRIDE;                 // value index = (row-in-unit, re/im)
                const int v = (c * PC + (j >> 1)) * VPP + gv;
                s_img[wave][(i * VP + v) * 2 + (j & 1)] = val[0];
            }
            // refill the slot just consumed, D units ahead (crossing into the next group)
            if (ui + D < UPG) load_unit(u, g_cur, ui + D);
            else load_unit(u, nxt, ui + D - UPG);
        }
        // per-lane inputs of the next group: in flight during this group's core
        in_nx = Core::load(d, A, nxt * EPW + lane / VP, v_mine, nxt * EPW + lane / VP < d.E);

        // the wave's own LDS writes -> its own reads (LDS is in-order per wave; no other wave
        // touches this slice); the barrier only pins the compiler's ordering
        __builtin_amdgcn_wave_barrier();
        const float2 img = *reinterpret_cast<const float2*>(&s_img[wave][lane * 2]);
        Core::template run<VP>(d, P, A, e_mine, v_mine, active, img, in);
        __builtin_amdgcn_wave_barrier();
    };

    In inA = Core::load(d, A, grp * EPW + lane / VP, v_mine, grp * EPW + lane / VP < d.E), inB;
    while (true) {
        do_group(grp, inA, inB);
        grp += stride;
        if (grp >= grp_end) break;
        do_group(grp, inB, inA);
        grp += stride;
        if (grp >= grp_end) break;
    }
}

// Waves per CU to launch (each strides over groups) and ring depth.  Measured on MI355X at
// C3 (E=32768, V=8, M=64), ring depth D x waves/CU -> us per step:
//   D=1: 26.5 (8)  27.2 (12)  26.6 (16)      D=2: 27.0 (8)  26.7 (12)  26.3 (16)
//   D=4: 28.0 (4)  27.7 (8)   28.0 (12)      D=8: 29.7 (4)  29.7 (8)   29.6 (12)
// i.e. a shallow ring with more resident waves wins: deeper rings cost registers (fewer
// waves) and buy nothing once ~8 waves/CU already keep >= 16 KiB in flight each.
static int pipe_waves_per_cu() {
    static int v = [] {
        const char* s = std::getenv("RISVEC_PIPE_WAVES_PER_CU");
        const int x = s ? std::atoi(s) : 0;
        return (x >= 1 && x <= 32) ? x : 8;
    }();
    return v;
}

static int env_int(const char* name, int dflt) {
    const char* s = std::getenv(name);
    return s ? std::atoi(s) : dflt;
}

template <int V, int M, int D, class Core>
static hipError_t launch_pipe(const RisVecState& s, const typename Core::Params& p, const typename Core::Args& a,
                              hipStream_t st) {
    using S = PipeShape<V, M>;
    const int n_groups = (s.n_envs + S::EPW - 1) / S::EPW;
    const int wpb = kBlock / kWave;
    long long want_waves = (long long)num_cus() * pipe_waves_per_cu();
    if (want_waves > n_groups) want_waves = n_groups;
    // balance: every wave gets the same number of groups (the last one possibly fewer)
    const long long per_wave = (n_groups + want_waves - 1) / want_waves;
    want_waves = (n_groups + per_wave - 1) / per_wave;
    const unsigned grid = (unsigned)((want_waves + wpb - 1) / wpb);
    // Non-temporal h_r / theta loads once the per-step stream no longer fits the 256 MiB Infinity Cache.  Measured
    // (same box, E x 8 x 64, us per step default / nt): 40 960 envs (188 MiB) 32.9 / 39.0, 57 344 (263 MiB) 46.0 / 53.8,
    // 65 536 (288 MiB) 59.4 / 57.8, 262 144 (1.15 GiB) 245-250 / 221-223; 32 768 x 16 x 256 (1.1 GiB) with the BCD sweep
    // 276-284 / 240-252 -- below the cache size the re-read of last step's lines is worth more than the hint, above it
    // the hint is worth 10 %.  RISVEC_PIPE_NT = 0 / 1 force it off / on; RISVEC_PIPE_CHUNKED (1: contiguous group range
    // per wavefront, 2: per workgroup) is an experiment that lost at every size but one (214 vs 222 us at 262 144 envs).
    static const int chunked = env_int("RISVEC_PIPE_CHUNKED", 0);
    static const int nt_mode = env_int("RISVEC_PIPE_NT", 2);              // 0 never, 1 always, 2 by stream size
    const long long stream_bytes = (long long)s.n_envs * (8LL * V * M + 8LL * M);
    const bool nt = nt_mode == 1 || (nt_mode == 2 && stream_bytes > tuning().pipe_nt_from);    // 1.055 x the Infinity Cache
    const int pw = chunked == 1 ? (int)per_wave : (chunked == 2 ? -(int)per_wave : 0);
    if (nt) hipLaunchKernelGGL((k_step_fused_pipe<V, M, D, Core, true>), dim3(grid), dim3(kBlock), 0, st, dims_of(s), p, a, n_groups, pw);
    else hipLaunchKernelGGL((k_step_fused_pipe<V, M, D, Core, false>), dim3(grid), dim3(kBlock), 0, st, dims_of(s), p, a, n_groups, pw);
    note_kernel("k_step_fused_pipe<%d,%d,%d,%s%s>", V, M, D, Core::name(), nt ? ",NT" : "");
    return hipGetLastError();
}

static bool pipe_disabled() {
    static const bool off = std::getenv("RISVEC_NO_PIPE") != nullptr;    // A/B switch for experiments
    return off;
}

// compile-time shapes shared by the three cores
template <class Core>
static hipError_t dispatch_pipe(const RisVecState& s, const typename Core::Params& p, const typename Core::Args& a,
                                hipStream_t st) {
    if (pipe_disabled()) return hipErrorNotSupported;
    const int V = s.n_veh, M = s.n_ris;
    if (V == 8 && M == 64) return launch_pipe<8, 64, 2, Core>(s, p, a, st);
    if (V == 8 && M == 36) return launch_pipe<8, 36, 2, Core>(s, p, a, st);
    if (V == 8 && M == 40) return launch_pipe<8, 40, 2, Core>(s, p, a, st);
    if (V == 4 && M == 16) return launch_pipe<4, 16, 4, Core>(s, p, a, st);
    if (V == 16 && M == 64) return launch_pipe<16, 64, 2, Core>(s, p, a, st);
    if (V == 16 && M == 256) return launch_pipe<16, 256, 2, Core>(s, p, a, st);
    return hipErrorNotSupported;
}

hipError_t launch_step_fused_pipe(const RisVecState& s, const RisVecParams& p, const StepArgs& a,
                                  hipStream_t st) {
    if (!pipe_disabled() && s.n_veh == 8 && s.n_ris == 64) {             // ring-depth experiment knob
        static const int depth = [] { const char* e = std::getenv("RISVEC_PIPE_DEPTH"); return e ? std::atoi(e) : 2; }();
        if (depth == 1) return launch_pipe<8, 64, 1, MarlCore>(s, p, a, st);
        if (depth == 4) return launch_pipe<8, 64, 4, MarlCore>(s, p, a, st);
        if (depth == 8) return launch_pipe<8, 64, 8, MarlCore>(s, p, a, st);
    }
    return dispatch_pipe<MarlCore>(s, p, a, st);
}

hipError_t launch_step_fused_pipe_ring(const RisVecState& s, const RisVecParams& p, const StepArgs& a, hipStream_t st) {
    if (pipe_disabled()) return hipErrorNotSupported;
    const int V = s.n_veh, M = s.n_ris;
    if (V == 8 && M == 64) return launch_pipe<8, 64, 2, MarlRingCore<8>>(s, p, a, st);
    if (V == 8 && M == 36) return launch_pipe<8, 36, 2, MarlRingCore<8>>(s, p, a, st);
    if (V == 8 && M == 40) return launch_pipe<8, 40, 2, MarlRingCore<8>>(s, p, a, st);
    if (V == 4 && M == 16) return launch_pipe<4, 16, 4, MarlRingCore<4>>(s, p, a, st);
    if (V == 16 && M == 64) return launch_pipe<16, 64, 2, MarlRingCore<16>>(s, p, a, st);
    if (V == 16 && M == 256) return launch_pipe<16, 256, 2, MarlRingCore<16>>(s, p, a, st);
    return hipErrorNotSupported;
}

hipError_t launch_sarl_pipe(const RisVecState& s, const RisVecSarlParams& p, const SarlArgs& a, hipStream_t st) {
    return dispatch_pipe<SarlCore>(s, p, a, st);
}

hipError_t launch_gain_pipe(const RisVecState& s, hipStream_t st) {
    const GainArgs a{s.pl, s.h_r, s.theta, s.b, s.h_d, s.gain};
    return dispatch_pipe<GainCore>(s, 0, a, st);
}

}  // namespace risvec
