// f1: the single-agent environment variant, Simulation-SARL/Environment.py
// (SENV): step(action_power, action_phase) SENV:321-359 with compute_data_rate SENV:149-171
// and get_next_phase SENV:133-139.  The phases come from the agent instead of a BCD sweep;
// the rate is a natural log against sigma^2 = 1e-14; local processing follows the cube-root
// CPU model; the reward is power + buffer length with two penalties.
//
// Same work decomposition as the generic fused MARL kernel (k_step.hip): a wave owns 64/VP
// envs, streams their h_r rows through the shared cascade, parks the reduced sums in its LDS
// slice, then one lane per (env, vehicle) runs the step.  theta = exp(j angle) is written by
// k_set_phase on the same stream just before (risvec_sarl_step launches both).
#include "risvec_sarl.hpp"

namespace risvec {

template <int VP, int G, int VEC>
__global__ void __launch_bounds__(kBlock)
k_sarl_step(Dims d, RisVecSarlParams P, SarlArgs A) {
    constexpr int EPW = kWave / VP;
    constexpr int VPP = kWave / G;
    __shared__ float2 s_img[kBlock / kWave][kWave];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int gl = lane % G, gv = lane / G;
    const int e0 = (blockIdx.x * (kBlock / kWave) + wave) * EPW;
    const int V = d.V, M = d.M;
    for (int i = 0; i < EPW; ++i) {
        const int e = e0 + i;                              // wave-uniform
        if (e >= d.E) break;
        const float* trow = A.theta + (long long)e * M * 2;
        for (int v0 = 0; v0 < V; v0 += VPP) {
            const int v = v0 + gv;
            const bool valid = v < V;
            const float* hrow = A.h_r + ((long long)e * V + (valid ? v : 0)) * M * 2;
            const float2 img = cascade_row<G, VEC>(hrow, trow, A.b, M, gl, valid);
            if (valid && gl == 0) s_img[wave][i * VP + v] = img;
        }
    }
    __syncthreads();
    const int e_mine = e0 + lane / VP, v_mine = lane % VP;
    const bool active = e_mine < d.E && v_mine < V;
    float g = 0.f, p0 = 0.f, p1 = 0.f, B = 0.f;
    if (active) {
        const long long idx = (long long)e_mine * V + v_mine;
        g = gain_from_img(s_img[wave][lane], A.pl[idx], nullptr, idx);
        A.gain[idx] = g;
        p0 = A.action_power[(long long)e_mine * 2 * V + v_mine];
        p1 = A.action_power[(long long)e_mine * 2 * V + V + v_mine];
        B = A.data_buf[idx];
    }
    sarl_core<VP>(d, P, A, e_mine, v_mine, active, g, p0, p1, B);
}

template <int VP>
static hipError_t launch_sarl_vp(const RisVecState& s, const RisVecSarlParams& p, const SarlArgs& a, hipStream_t st) {
    const long long threads = (long long)s.n_envs * VP;
    const unsigned grid = (unsigned)((threads + kBlock - 1) / kBlock);
    const Dims d = dims_of(s);
    const bool even = (s.n_ris & 1) == 0;
    const int g = pick_group(s.n_ris, even ? 2 : 1, VP);
#define RISVEC_SARL(GG)                                                                           \
    if (g == GG) {                                                                                \
        if (even) hipLaunchKernelGGL((k_sarl_step<VP, GG, 2>), dim3(grid), dim3(kBlock), 0, st, d, p, a); \
        else hipLaunchKernelGGL((k_sarl_step<VP, GG, 1>), dim3(grid), dim3(kBlock), 0, st, d, p, a);      \
        return hipGetLastError();                                                                 \
    }
    if constexpr (kWave / VP <= 8) { RISVEC_SARL(8) }
    if constexpr (kWave / VP <= 16) { RISVEC_SARL(16) }
    if constexpr (kWave / VP <= 32) { RISVEC_SARL(32) }
    RISVEC_SARL(64)
#undef RISVEC_SARL
    return hipErrorInvalidValue;
}

hipError_t launch_sarl_step(const RisVecState& s, const RisVecSarlParams& p, const float* action_power,
                            const float* action_phase, const int32_t* arrivals, uint64_t seed,
                            uint32_t counter, uint32_t flags, hipStream_t st) {
    if (action_phase) {                                    // get_next_phase, SENV:133-139
        const hipError_t err = launch_set_phase(s, action_phase, st);
        if (err != hipSuccess) return err;
    }
    SarlArgs a;
    a.action_power = action_power; a.arrivals = arrivals; a.pl = s.pl; a.h_r = s.h_r; a.theta = s.theta;
    a.b = s.b; a.gain = s.gain; a.data_buf = s.data_buf; a.rate = s.rate; a.data_t = s.data_t;
    a.data_p = s.data_p; a.reward = s.reward; a.over_power = s.over_power; a.over_data = s.over_data;
    a.obs = s.obs; a.metrics = s.metrics; a.seed = seed; a.counter = counter; a.flags = flags;
    {
        const hipError_t err = launch_sarl_pipe(s, p, a, st);
        if (err != hipErrorNotSupported) return err;
    }
    switch (pow2_ceil(s.n_veh)) {
        case 1: return launch_sarl_vp<1>(s, p, a, st);
        case 2: return launch_sarl_vp<2>(s, p, a, st);
        case 4: return launch_sarl_vp<4>(s, p, a, st);
        case 8: return launch_sarl_vp<8>(s, p, a, st);
        case 16: return launch_sarl_vp<16>(s, p, a, st);
        case 32: return launch_sarl_vp<32>(s, p, a, st);
        case 64: return launch_sarl_vp<64>(s, p, a, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace risvec
