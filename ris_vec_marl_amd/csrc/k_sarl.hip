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
#include "risvec_step.hpp"

namespace risvec {

struct SarlArgs {
    const float* action_power;   // [E,2,V]
    const int32_t* arrivals;     // [E,V] or nullptr
    const float* pl;
    const float* h_r;
    const float* theta;
    const float* b;
    float* gain;
    float* data_buf;
    float* rate;
    float* data_t;
    float* data_p;
    float* reward;
    float* over_power;
    float* over_data;
    float* obs;
    float* metrics;
    uint64_t seed;
    uint32_t counter;
    uint32_t flags;
};

template <int VP>
__device__ __forceinline__ void sarl_core(const Dims& d, const RisVecSarlParams& P, const SarlArgs& A, int e,
                                          int v, bool active, float gain) {
    const int V = d.V;
    const long long idx = (long long)e * V + v;
    float p0 = 0.f, p1 = 0.f, B = 0.f;
    if (active) {
        p0 = A.action_power[(long long)e * 2 * V + v];
        p1 = A.action_power[(long long)e * 2 * V + V + v];
        B = A.data_buf[idx];
    }
    const float tf = P.time_fast;
    // SENV:159  rate = ln(1 + p0 |cascaded_gain|^2 / sigma^2),  sigma = 1e-7
    const float rate = log2_1p(p0 * gain * 1.0e14f) * 0.6931471805599453f;
    const float data_t = rate * tf * P.bandwidth_mhz * 1000.0f;              // SENV:329
    const float data_p = fdiv(fdiv(cbrtf(fdiv(p1, P.k_cpu)) * tf, P.cycles_l), 1000.0f);   // SENV:330
    float Bn = B - (data_t + data_p);                                        // SENV:333
    const bool neg = Bn < 0.f;
    const float need = fmaxf(0.f, Bn + data_p);                              // SENV:336
    const float x = fdiv(need * 1000.0f * P.cycles_l, tf);                   // SENV:318-319
    const float proc_rev = x * x * x * P.k_cpu;
    const float over_power = neg ? p1 - proc_rev : 0.f;
    const float over_data = neg ? -Bn : 0.f;                                 // SENV:337, 340
    Bn = neg ? 0.f : Bn;                                                     // SENV:338
    const float base = -(P.t_factor1 * (p0 + p1)) - P.t_factor2 * Bn;        // SENV:344-352
    const float rew = Bn > 0.f ? base - P.penalty1 : (over_data > 2.f ? base - P.penalty2 : base);
    int arr = 0;                                                             // SENV:354-356
    if (A.arrivals) {
        if (active) arr = A.arrivals[idx];
    } else {
        const uint4 r = philox4x32_10((uint32_t)(d.env_offset + e), (uint32_t)v, A.counter, kSiteArrivals, A.seed);
        arr = poisson_from_u(u01(r.x), P.poisson_cdf);
    }
    const float Bo = Bn + (float)arr * tf * 1000.0f;
    const float rew_sum = gsum<VP>(active ? rew : 0.f);
    if (active) {
        A.data_buf[idx] = Bo;
        A.rate[idx] = rate;
        A.data_t[idx] = data_t;
        A.data_p[idx] = data_p;
        A.reward[idx] = rew;
        A.over_power[idx] = over_power;
        A.over_data[idx] = over_data;
        if (A.flags & RISVEC_STEP_OBS) {
            // tail of ddpg_train.py:47-73 (the theta slice in front of it is the agent's own action)
            float* o = A.obs + idx * 5;
            o[0] = Bo * 0.1f; o[1] = data_t * 0.1f; o[2] = data_p * 0.1f; o[3] = over_data * 0.1f; o[4] = rate * 0.05f;
        }
        if (v == 0) A.metrics[(long long)e * RISVEC_METRICS] = rew_sum * __builtin_amdgcn_rcpf((float)V);   // SENV:358
    }
}

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
    float g = 0.f;
    if (active) {
        const long long idx = (long long)e_mine * V + v_mine;
        g = gain_from_img(s_img[wave][lane], A.pl[idx], nullptr, idx);
        A.gain[idx] = g;
    }
    sarl_core<VP>(d, P, A, e_mine, v_mine, active, g);
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
