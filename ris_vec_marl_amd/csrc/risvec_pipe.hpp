// Pieces shared by the software-pipelined kernels (k_step_pipe.hip) and the latency-shaped /
// multi-step kernels (k_step_lat.hip): the compile-time tiling of one env group, the register
// image of a load unit, the transposing butterfly and the MARL core.
#pragma once

#include <cstdlib>
#include <cstring>

#include "risvec_step.hpp"

namespace risvec {

// Tiling of one env onto a wavefront.  G lanes share one (env, vehicle) row of h_r (16-byte loads: two complex
// elements per lane), NIT loads per lane cover the row, 64 / G rows are read per pass.  MC = the compile-time RIS
// size, or 0: any EVEN M with ceil(M / 2 / G) == NIT, read from Dims at run time (round 3: the reference's own
// RIS-element study runs M = 20 ... 120, plt/plt-ris.py:7).
template <int V_, int G_, int NIT_, int MC_>
struct FusedShape {
    static constexpr int V = V_, VP = V_, G = G_, NIT = NIT_, MC = MC_;
    static_assert(pow2_ceil(V_) == V_, "fused fast kernels are instantiated for power-of-two V");
    static_assert(MC_ % 2 == 0, "fused fast kernels need an even M (16-byte loads)");
    static constexpr int EPW = kWave / VP;                     // envs per group
    static constexpr int VPP = kWave / G;                      // rows per pass
    static constexpr int PASSES = V / VPP;
    static_assert(PASSES >= 1 && PASSES * VPP == V, "rows per pass must divide V");
    static constexpr int PC = PASSES >= 4 ? 4 : PASSES;        // rows per unit per lane-group
    static constexpr int CHUNKS = PASSES / PC;
    static_assert(CHUNKS * PC == PASSES, "PASSES must be a multiple of PC");
    static constexpr int UPG = EPW * CHUNKS;                   // units per group
    static constexpr int K = 2 * PC;                           // values reduced per unit
    static constexpr int WSTRIDE = G / K;                      // writer lanes: gl % WSTRIDE == 0
    static_assert(G >= K, "need at least K lanes per row");
    static constexpr bool FIXED = MC_ != 0;
    static constexpr bool RAGGED = !FIXED || ((MC_ / 2) % G_ != 0);     // some lanes lie past the end of a row
    // complex pairs per row
    static __device__ __forceinline__ int np(const Dims& d) { return FIXED ? MC_ / 2 : (d.M >> 1); }
};

// lanes per row / loads per lane for an even M (host and device): enough lanes to cover M / 2 pairs in one pass when
// possible, but never more rows per pass than an env has, and never fewer than 8 lanes
__host__ __device__ constexpr int fused_g(int V, int M) {
    const int np = M / 2;
    const int g0 = pow2_ceil(np) > kWave ? kWave : pow2_ceil(np);
    const int gmin = (kWave / pow2_ceil(V)) < 8 ? 8 : (kWave / pow2_ceil(V));
    return g0 < gmin ? gmin : g0;
}
__host__ __device__ constexpr int fused_nit(int V, int M) { return (M / 2 + fused_g(V, M) - 1) / fused_g(V, M); }

template <int V, int M>
struct PipeShape : FusedShape<V, fused_g(V, M), fused_nit(V, M), M> {
    static constexpr int NP = M / 2;                           // complex pairs per row
};

template <int PC, int NIT>
struct Unit {
    float4 h[PC][NIT];
    float4 t[NIT];
};

// ---------------------------------------------------------------------------
// What runs on the reduced cascade sums: the pipeline is the same for the MARL step, the SARL
// step and the gain-only kernel; a Core supplies its parameter / argument types, the per-lane
// inputs it wants prefetched one group ahead, and the per-lane work.
// ---------------------------------------------------------------------------
struct MarlCore {
    static const char* name() { return "MarlCore"; }
    using Params = RisVecParams;
    using Args = StepArgs;
    using In = StepIn;
    static __device__ __forceinline__ In load(const Dims& d, const Args& A, int e, int v, bool active) {
        return load_step_in(d, A, e, v, active);
    }
    static __device__ __forceinline__ void hold(const In& in) {
        asm volatile("" ::"v"(in.a0), "v"(in.a1), "v"(in.B), "v"(in.Q0), "v"(in.pl), "v"(in.part), "v"(in.G), "v"(in.arr_in));
    }
    template <int VP>
    static __device__ __forceinline__ void run(const Dims& d, const Params& P, const Args& A, int e, int v,
                                               bool active, float2 img, const In& in) {
        float g = 0.f;
        if (active) {
            const long long idx = (long long)e * d.V + v;
            g = gain_from_img(img, in.pl, A.h_d, idx);
            A.gain[idx] = g;
        }
        step_core<VP>(d, P, A, e, v, active, g, in);
    }
};

// ---------------------------------------------------------------------------
// Dispatch thresholds, derived ONCE from the device (round 3; they were literals tuned on one box).
//   cus            hipDeviceProp_t::multiProcessorCount
//   ic_bytes       the Infinity Cache (memory-side L3) of the architecture: not in hipDeviceProp_t, so a table by
//                  gcnArchName (gfx950 / gfx942: 256 MiB)
//   pipe_nt_from   h_r + theta bytes of one step from which the software pipeline reads them with the non-temporal
//                  hint = 1.055 x ic_bytes (measured crossover 263 ... 288 MiB on a 256 MiB cache: below it last
//                  step's lines are still resident and the hint throws that away)
//   lat_nt_from    from here the latency-shaped kernel streams with the non-temporal hint = 1.29 x ic_bytes (measured
//                  crossover 294 ... 368 MiB); between ic_bytes and this it reads with the default policy and walks the
//                  envs in alternating directions from step to step (round 3: 3-12 % faster than the hint in that band,
//                  equal at 1.4 x, 4 % slower at 5 x)
// The RISVEC_* environment switches remain as overrides for same-box A/Bs and the bit-identity tests.
// ---------------------------------------------------------------------------
struct Tuning {
    int cus;
    long long ic_bytes, pipe_nt_from, lat_nt_from, colsum_nt_from;
};

inline const Tuning& tuning() {
    static const Tuning t = [] {
        Tuning r{256, 256LL << 20, 0, 0, 0};
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, dev) == hipSuccess) {
                if (prop.multiProcessorCount > 0) r.cus = prop.multiProcessorCount;
                const char* a = prop.gcnArchName;
                if (std::strncmp(a, "gfx90a", 6) == 0) r.ic_bytes = 0;              // MI200: no memory-side cache
                else if (std::strncmp(a, "gfx94", 5) == 0 || std::strncmp(a, "gfx95", 5) == 0) r.ic_bytes = 256LL << 20;
            }
        }
        auto mb = [](const char* name, long long dflt) {
            const char* e = std::getenv(name);
            return e ? (std::atoll(e) << 20) : dflt;
        };
        r.pipe_nt_from = mb("RISVEC_PIPE_NT_MB", r.ic_bytes + r.ic_bytes * 55 / 1000);
        r.lat_nt_from = mb("RISVEC_LAT_NT_MB", r.ic_bytes + r.ic_bytes * 29 / 100);
        r.colsum_nt_from = r.ic_bytes + r.ic_bytes * 55 / 1000;
        return r;
    }();
    return t;
}

inline int num_cus() { return tuning().cus; }

// MarlCore + the transition store (StepArgs::ring): the ring's per-lane inputs (the observation being replaced, the
// agent's partner-probability row, its mask row) are prefetched one group ahead with the step inputs.
template <int VPC>
struct MarlRingCore {
    static const char* name() { return "MarlCore+ring"; }
    using Params = RisVecParams;
    using Args = StepArgs;
    struct In { StepIn s; RingIn<VPC> r; };
    static __device__ __forceinline__ In load(const Dims& d, const Args& A, int e, int v, bool active) {
        return In{load_step_in(d, A, e, v, active), load_ring_in<VPC>(d, A, e, v, active)};
    }
    static __device__ __forceinline__ void hold(const In& in) {
        MarlCore::hold(in.s);
#pragma unroll
        for (int k = 0; k < 5; ++k) asm volatile("" ::"v"(in.r.so[k]));
#pragma unroll
        for (int k = 0; k < VPC; ++k) asm volatile("" ::"v"(in.r.prow[k]));
#pragma unroll
        for (int k = 0; k < VPC / 4; ++k) asm volatile("" ::"v"(in.r.mk[k]));
    }
    template <int VP>
    static __device__ __forceinline__ void run(const Dims& d, const Params& P, const Args& A, int e, int v,
                                               bool active, float2 img, const In& in) {
        static_assert(VP == VPC, "ring core instantiated for another V");
        float g = 0.f;
        if (active) {
            const long long idx = (long long)e * d.V + v;
            g = gain_from_img(img, in.s.pl, A.h_d, idx);
            A.gain[idx] = g;
        }
        step_core<VP, false, false, RingIn<VPC>>(d, P, A, e, v, active, g, in.s, nullptr, &in.r);
    }
};

}  // namespace risvec
