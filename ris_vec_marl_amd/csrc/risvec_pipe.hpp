// Pieces shared by the software-pipelined kernels (k_step_pipe.hip) and the latency-shaped /
// multi-step kernels (k_step_lat.hip): the compile-time tiling of one env group, the register
// image of a load unit, the transposing butterfly and the MARL core.
#pragma once

#include <cstdlib>

#include "risvec_step.hpp"

namespace risvec {

template <int V, int M>
struct PipeShape {
    static constexpr int VP = pow2_ceil(V);
    static_assert(V == VP, "pipelined kernels are instantiated for power-of-two V");
    static_assert(M % 2 == 0, "pipelined kernels need an even M (16-byte loads)");
    static constexpr int EPW = kWave / VP;                     // envs per group
    static constexpr int NP = M / 2;                           // complex pairs per row
    static constexpr int G0 = pow2_ceil(NP) > kWave ? kWave : pow2_ceil(NP);
    static constexpr int GMIN = (kWave / VP) < 8 ? 8 : (kWave / VP);
    static constexpr int G = G0 < GMIN ? GMIN : G0;            // lanes per row
    static constexpr int NIT = (NP + G - 1) / G;               // 16-B loads per lane per row
    static constexpr int VPP = kWave / G;                      // rows per pass
    static constexpr int PASSES = V / VPP;
    static_assert(PASSES * VPP == V, "rows per pass must divide V");
    static constexpr int PC = PASSES >= 4 ? 4 : PASSES;        // rows per unit per lane-group
    static constexpr int CHUNKS = PASSES / PC;
    static_assert(CHUNKS * PC == PASSES, "PASSES must be a multiple of PC");
    static constexpr int UPG = EPW * CHUNKS;                   // units per group
    static constexpr int K = 2 * PC;                           // values reduced per unit
    static constexpr int WSTRIDE = G / K;                      // writer lanes: gl % WSTRIDE == 0
    static_assert(G >= K, "need at least K lanes per row");
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

inline int num_cus() {
    static int n = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                cus = prop.multiProcessorCount;
        }
        return cus;
    }();
    return n;
}

}  // namespace risvec
