// SARL step core (Simulation-SARL/Environment.py:321-359) shared by the generic kernel
// (k_sarl.hip) and the software-pipelined one (k_step_pipe.hip).
#pragma once

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

// p0 / p1 = offload / local power of this lane, B = its DataBuf (0 for inactive lanes)
template <int VP>
__device__ __forceinline__ void sarl_core(const Dims& d, const RisVecSarlParams& P, const SarlArgs& A, int e,
                                          int v, bool active, float gain, float p0, float p1, float B) {
    const int V = d.V;
    const long long idx = (long long)e * V + v;
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

// pipelined fused kernels for compile-time shapes (k_step_pipe.hip); hipErrorNotSupported otherwise
hipError_t launch_sarl_pipe(const RisVecState& s, const RisVecSarlParams& p, const SarlArgs& a, hipStream_t st);

}  // namespace risvec
