// The hot path: RIS cascaded gain (K3), step() (K4) and their fusion (K34).
//
// Reference: Simulation-MARL-BCD/Environment.py (ENV): update_channel_gains "free"
// ENV:263-273, compute_data_rate ENV:331-372, step ENV:547-731; observation
// marl_train_bcd.py:819-827, action map marl_train_bcd.py:1601-1608.
//
// Work decomposition (wave64, no MFMA: this is a bandwidth-bound reduction path):
//   * cascade: a group of G lanes (G = 8..64, power of two) owns one (env, vehicle)
//     row h_r[e,v,:], reads it with 16-byte loads (two complex per lane), multiplies
//     by w = theta[e,:] * b[:] and reduces the complex partial sums over the group
//     with wavefront shuffles.
//   * step: one lane per (env, vehicle); the V lanes of an env sit in an aligned group
//     of VP = pow2ceil(V) lanes, so the four cross-vehicle couplings of step() (NOMA
//     partner gain/power, sum of edge cycles, the V-means) are group shuffles.
//   * fused: a wave owns 64/VP consecutive envs; it streams their h_r rows, parks the
//     reduced sums in its private LDS slice, then every lane picks up "its" (env,
//     vehicle) sum and runs the step.  One HBM pass over h_r/theta, gains never
//     round-trip through HBM before use.
#include "risvec_launch.hpp"

namespace risvec {

struct StepArgs {
    const float* action;
    const int32_t* partner;
    const int32_t* n_groups;
    const int32_t* arrivals;
    const float* pl;
    const float* h_r;
    const float* theta;
    const float* b;
    const float* h_d;
    float* gain;
    float* data_buf;
    float* mec_q;
    float* rate;
    float* data_t;
    float* data_p;
    float* reward;
    float* over_power;
    float* obs;
    float* metrics;
    float* power_w;
    uint64_t seed;
    uint32_t counter;
    uint32_t flags;
};

// compute_data_rate (ENV:331-372) for one lane; all lanes of the VP-group must call.
// near = u1 if gain1 > gain2 else u2 (ENV:355-360): a vehicle listed second is "near" on ties.
template <int VP>
__device__ __forceinline__ float noma_rate(const RisVecParams& P, float pw0, float gain, int part, int G) {
    const int lane = threadIdx.x & (kWave - 1);
    const int base = lane & ~(VP - 1);
    const bool pair = part >= 0, single = part == RISVEC_PARTNER_SINGLE;
    const bool second = part >= RISVEC_PARTNER_SECOND;
    const int src = base + (pair ? (part & (VP - 1)) : (lane - base));
    const float g_p = __shfl(gain, src, kWave);
    const float pw_p = __shfl(pw0, src, kWave);
    const bool near = second ? !(g_p > gain) : (gain > g_p);
    const float sig = pw0 * gain;                                            // ENV:347, 362, 367
    const float den = (pair && !near) ? (pw_p * gain + P.noise_power) : P.noise_power;   // ENV:363-364
    const float sinr = sig / den;
    const float frac = 1.0f / (float)max(1, G);                              // ENV:341-342
    const float rate = frac * (log1pf(sinr) * 1.4426950408889634f);         // log2(1 + sinr)
    return (pair || single) ? rate : 0.f;
}

// ---------------------------------------------------------------------------
// step() for one (env, vehicle) lane.  Called by ALL 64 lanes (shuffles inside);
// `active` masks lanes beyond V or E.
// ---------------------------------------------------------------------------
template <int VP>
__device__ __forceinline__ void step_core(const Dims& d, const RisVecParams& P, const StepArgs& A,
                                          int e, int v, bool active, float gain) {
    const int V = d.V;
    const long long idx = (long long)e * V + v;
    const float eps = 1e-12f;

    float a0 = 0.f, a1 = 0.f, B = 0.f, Q0 = 0.f;
    int part = RISVEC_PARTNER_NONE, G = 1;
    if (active) {
        if (A.flags & RISVEC_STEP_POLICY_ACTION) {
            // marl_train_bcd.py:1601-1608: [-1,1] -> [0,1], CPU share floored
            const float2 pa = *reinterpret_cast<const float2*>(A.action + idx * 2);
            a0 = (fminf(fmaxf(pa.x, -0.999f), 0.999f) + 1.f) * 0.5f;
            a1 = (fminf(fmaxf(pa.y, -0.999f), 0.999f) + 1.f) * 0.5f;
        } else {
            a0 = A.action[(long long)e * 2 * V + v];
            a1 = A.action[(long long)e * 2 * V + V + v];
        }
        B = A.data_buf[idx];
        part = A.partner[idx];
        G = A.n_groups[e];
        Q0 = A.mec_q[e];
    }
    // ENV:574-577
    float fl = P.cpu_share_floor;
    if (!isfinite(fl)) fl = 0.10f;
    fl = fmaxf(0.f, fminf(fl, 0.95f));
    if (A.flags & RISVEC_STEP_POLICY_ACTION) a1 = fmaxf(a1, fl);

    // (1) power projection, ENV:555-561
    float c0 = fmaxf(a0, 0.f) * P.power_scale;
    float c1 = fmaxf(a1, 0.f) * P.power_scale;
    const float s = c0 + c1;
    if (s > 1.f) {
        const float den = s + 1e-12f;
        c0 = c0 / den;
        c1 = c1 / den;
    }
    const float pw0 = c0 * P.p_max, pw1 = c1 * P.p_max;

    // (2) rate, ENV:331-372
    const float rate = noma_rate<VP>(P, pw0, gain, part, G);
    const float tf = P.time_fast, bw = P.bandwidth_mhz;
    const float data_t = rate * tf * bw * 1000.0f;                          // ENV:570

    // (3) cpu share, ENV:572-580
    const float cpu = fmaxf(fminf(fmaxf(a1, 0.f), 1.f), fl);
    const float f = cpu * P.f_local_max;
    const float Cpb = P.cycles_per_bit;

    // (4) local processing, ENV:585-592
    // When the CPU can clear the whole backlog, data_p = bc / (Cpb*1000) equals B up to
    // rounding (1e-16 in the float64 reference).  In float32 that rounding (1e-7 B) would
    // leak into `rem`, `off` and t_tx = off / throughput, so the identity is used directly.
    const float bc = B * 1000.0f * Cpb;
    const float cap = f * tf;
    const bool clears = cap >= bc;
    const float used = clears ? bc : cap;
    const float data_p = clears ? B : cap / (Cpb * 1000.0f);

    // (5) offload, ENV:595-601
    const float rem = fmaxf(0.f, B - data_p);
    const float off = fminf(data_t, rem);
    const float thr = rate * bw * 1000.0f;
    const float t_tx = off / (thr + 1e-12f);

    // (6) MEC queue, ENV:604-610
    const float ein = off * 1000.0f * Cpb;
    const float ein_sum = group_sum<VP>(active ? ein : 0.f);
    float Q = Q0 + ein_sum;
    const float edge_cap = P.f_edge_max * tf;
    const float svc = fminf(edge_cap, Q);
    Q -= svc;

    // (7) backlog, ENV:617-618
    float Bn = fmaxf(0.f, B - (data_p + off));

    // (8) delays, ENV:622-633
    const float d_loc = fmaxf(0.f, bc - ein) / (f + eps);
    const float share = ein / (ein_sum + eps);
    const float d_q = share * (Q0 / (P.f_edge_max + eps));
    const float d_c = ein / (P.f_edge_max + eps);
    const float delay = d_loc + t_tx + d_q + d_c;

    // (9) energy, ENV:659-666
    const float E_tx = pw0 * t_tx;
    const float E_loc = P.k_cpu * (f * f) * used;
    const float energy = E_tx + E_loc;

    // (10) QoS, ENV:669-677
    const bool viol = P.qos_enable && ((rate < P.r_min_bpshz) || (delay > P.d_max_s));
    const float pen = viol ? P.qos_penalty : 0.f;

    // (11) reward, ENV:696-703
    const float cost = P.w_d * delay + P.w_e * energy;
    const float rew = fminf(fmaxf(-cost - pen, -P.reward_clip), P.reward_clip);

    // (12) arrivals, ENV:717-719
    int arr = 0;
    if (A.arrivals) {
        if (active) arr = A.arrivals[idx];
    } else {
        const uint4 r = philox4x32_10((uint32_t)(d.env_offset + e), (uint32_t)v, A.counter,
                                      kSiteArrivals, A.seed);
        arr = poisson_from_u(u01(r.x), P.poisson_cdf);
    }
    Bn += (float)arr * tf * 1000.0f;

    // (13) ENV:721-729
    const float over_power = fmaxf(0.f, (pw0 + pw1) - P.p_max);
    const float rew_sum = group_sum<VP>(active ? rew : 0.f);
    const float inv_v = 1.0f / (float)V;

    if (active) {
        A.data_buf[idx] = Bn;
        A.rate[idx] = rate;
        A.data_t[idx] = data_t;
        A.data_p[idx] = data_p;
        A.reward[idx] = rew;
        A.over_power[idx] = over_power;
        if (A.flags & RISVEC_STEP_OBS) {
            // marl_train_bcd.py:819-827 (element 3 = over_data/10 is always 0)
            float* o = A.obs + idx * 5;
            o[0] = Bn / 10.f; o[1] = data_t / 10.f; o[2] = data_p / 10.f; o[3] = 0.f; o[4] = rate / 20.f;
        }
        if (A.flags & RISVEC_STEP_POWER_W) {
            A.power_w[(long long)e * 2 * V + v] = E_tx / tf;
            A.power_w[(long long)e * 2 * V + V + v] = E_loc / tf;
        }
        if (v == 0) A.mec_q[e] = Q;
    }

    if (A.flags & RISVEC_STEP_METRICS) {
        const float z = 0.f;
        const float s_off = group_sum<VP>(active ? off : z);
        const float s_dp = group_sum<VP>(active ? data_p : z);
        const float s_b = group_sum<VP>(active ? B : z);
        const float s_dl = group_sum<VP>(active ? d_loc : z);
        const float s_dq = group_sum<VP>(active ? d_q : z);
        const float s_dc = group_sum<VP>(active ? d_c : z);
        const float s_tx = group_sum<VP>(active ? t_tx : z);
        const float s_ut = group_sum<VP>(active ? used / (cap + 1e-12f) : z);
        const float s_vi = group_sum<VP>(active && viol ? 1.f : z);
        const float s_de = group_sum<VP>(active ? delay : z);
        const float s_en = group_sum<VP>(active ? energy : z);
        if (active && v == 0) {
            float4* m = reinterpret_cast<float4*>(A.metrics + (long long)e * RISVEC_METRICS);
            m[0] = make_float4(rew_sum * inv_v, s_off, s_dp, Q);
            m[1] = make_float4(s_b * inv_v, s_dl * inv_v, s_dq * inv_v, s_dc * inv_v);
            m[2] = make_float4(s_tx * inv_v, svc / (edge_cap + 1e-12f), s_ut * inv_v, s_vi * inv_v);
            m[3] = make_float4(s_de * inv_v, s_en * inv_v, 0.f, 0.f);
        }
    } else if (active && v == 0) {
        A.metrics[(long long)e * RISVEC_METRICS] = rew_sum * inv_v;          // global_reward only
    }
}

// gain from the reduced cascade sum: | sqrt(pl) img + h_d |^2  (ENV:270-272; h_d = 0 there)
__device__ __forceinline__ float gain_from_img(float2 img, float pl, const float* h_d, long long idx) {
    if (h_d) {
        const float a = sqrtf(pl);
        const float2 hd = *reinterpret_cast<const float2*>(h_d + idx * 2);
        const float re = fmaf(a, img.x, hd.x), im = fmaf(a, img.y, hd.y);
        return re * re + im * im;
    }
    return pl * (img.x * img.x + img.y * img.y);
}

// ---------------------------------------------------------------------------
// cascade: sum_m theta[e,m] b[m] h_r[e,v,m] over a G-lane group.
// VEC = complex elements per load (2 -> float4, needs M even; 1 -> float2).
// ---------------------------------------------------------------------------
template <int G, int VEC>
__device__ __forceinline__ float2 cascade_row(const float* __restrict__ hrow,
                                              const float* __restrict__ trow,
                                              const float* __restrict__ b, int M, int gl, bool valid) {
    float2 acc = make_float2(0.f, 0.f);
    if (VEC == 2) {
        const int npair = M >> 1;
        for (int p = gl; p < npair; p += G) {
            if (valid) {
                const float4 h = *reinterpret_cast<const float4*>(hrow + 4 * p);
                const float4 t = *reinterpret_cast<const float4*>(trow + 4 * p);
                const float4 bb = *reinterpret_cast<const float4*>(b + 4 * p);
                const float2 w0 = cmul(make_float2(t.x, t.y), make_float2(bb.x, bb.y));
                const float2 w1 = cmul(make_float2(t.z, t.w), make_float2(bb.z, bb.w));
                acc = cfma(make_float2(h.x, h.y), w0, acc);
                acc = cfma(make_float2(h.z, h.w), w1, acc);
            }
        }
    } else {
        for (int m = gl; m < M; m += G) {
            if (valid) {
                const float2 h = *reinterpret_cast<const float2*>(hrow + 2 * m);
                const float2 t = *reinterpret_cast<const float2*>(trow + 2 * m);
                const float2 bb = *reinterpret_cast<const float2*>(b + 2 * m);
                acc = cfma(h, cmul(t, bb), acc);
            }
        }
    }
    acc.x = group_sum<G>(acc.x);
    acc.y = group_sum<G>(acc.y);
    return acc;
}

// K3: standalone gain kernel, one G-lane group per (env, vehicle)
template <int G, int VEC>
__global__ void __launch_bounds__(kBlock)
k_gain(Dims d, const float* __restrict__ h_r, const float* __restrict__ theta,
       const float* __restrict__ b, const float* __restrict__ pl, const float* __restrict__ h_d,
       float* __restrict__ gain) {
    constexpr int UPB = kBlock / G;                       // (env,veh) units per block
    const int gl = threadIdx.x % G;
    const long long u = (long long)blockIdx.x * UPB + threadIdx.x / G;
    const bool valid = u < (long long)d.E * d.V;
    const long long uu = valid ? u : 0;
    const long long e = uu / d.V;
    const float2 img = cascade_row<G, VEC>(h_r + uu * d.M * 2, theta + e * d.M * 2, b, d.M, gl, valid);
    if (valid && gl == 0) gain[u] = gain_from_img(img, pl[u], h_d, u);
}

// K4: standalone step kernel (cached gains; the reference's per-step cadence)
template <int VP>
__global__ void __launch_bounds__(kBlock)
k_step(Dims d, RisVecParams P, StepArgs A) {
    const long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
    const int e = (int)(t / VP), v = (int)(t % VP);
    const bool active = e < d.E && v < d.V;
    const float g = active ? A.gain[(long long)e * d.V + v] : 0.f;
    step_core<VP>(d, P, A, e, v, active, g);
}

// compute_data_rate as its own entry point (ENV:331-372)
template <int VP>
__global__ void __launch_bounds__(kBlock)
k_data_rate(Dims d, RisVecParams P, const float* __restrict__ p_off, const float* __restrict__ gain,
            const int32_t* __restrict__ partner, const int32_t* __restrict__ n_groups,
            float* __restrict__ rate_out) {
    const long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
    const int e = (int)(t / VP), v = (int)(t % VP);
    const bool active = e < d.E && v < d.V;
    const long long idx = (long long)e * d.V + v;
    const float g = active ? gain[idx] : 0.f;
    const float pw = active ? p_off[idx] : 0.f;
    const int part = active ? partner[idx] : RISVEC_PARTNER_NONE;
    const int G = active ? n_groups[e] : 1;
    const float r = noma_rate<VP>(P, pw, g, part, G);
    if (active) rate_out[idx] = r;
}

// K34: fused gain + step.  Each wave owns 64/VP consecutive envs.
template <int VP, int G, int VEC>
__global__ void __launch_bounds__(kBlock)
k_step_fused(Dims d, RisVecParams P, StepArgs A) {
    constexpr int EPW = kWave / VP;                        // envs per wave
    constexpr int VPP = kWave / G;                         // vehicles per pass
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

    const int e = e0 + lane / VP, v = lane % VP;
    const bool active = e < d.E && v < V;
    float g = 0.f;
    if (active) {
        const long long idx = (long long)e * V + v;
        g = gain_from_img(s_img[wave][lane], A.pl[idx], A.h_d, idx);
        A.gain[idx] = g;
    }
    step_core<VP>(d, P, A, e, v, active, g);
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
static StepArgs make_args(const RisVecState& s, const float* action, const int32_t* partner,
                          const int32_t* n_groups, const int32_t* arrivals, uint64_t seed,
                          uint32_t counter, uint32_t flags) {
    StepArgs a;
    a.action = action; a.partner = partner; a.n_groups = n_groups; a.arrivals = arrivals;
    a.pl = s.pl; a.h_r = s.h_r; a.theta = s.theta; a.b = s.b; a.h_d = s.h_d;
    a.gain = s.gain; a.data_buf = s.data_buf; a.mec_q = s.mec_q;
    a.rate = s.rate; a.data_t = s.data_t; a.data_p = s.data_p; a.reward = s.reward;
    a.over_power = s.over_power; a.obs = s.obs; a.metrics = s.metrics; a.power_w = s.power_w;
    a.seed = seed; a.counter = counter; a.flags = flags;
    return a;
}

// lanes per (env, vehicle) row: enough to cover M/VEC elements in one pass when
// possible, but never more vehicles per pass than an env has (keeps lanes busy).
static int pick_group(int M, int vec, int VP) {
    int g = pow2_ceil((M + vec - 1) / vec);
    if (g > kWave) g = kWave;
    const int gmin = kWave / VP;             // >= this => vehicles per pass <= VP
    if (g < gmin) g = gmin;
    if (g < 8) g = 8;
    return g;
}

template <int G>
static hipError_t launch_gain_g(const RisVecState& s, hipStream_t st) {
    const long long units = (long long)s.n_envs * s.n_veh;
    const unsigned grid = (unsigned)((units + (kBlock / G) - 1) / (kBlock / G));
    if ((s.n_ris & 1) == 0)
        hipLaunchKernelGGL((k_gain<G, 2>), dim3(grid), dim3(kBlock), 0, st, dims_of(s), s.h_r, s.theta,
                           s.b, s.pl, s.h_d, s.gain);
    else
        hipLaunchKernelGGL((k_gain<G, 1>), dim3(grid), dim3(kBlock), 0, st, dims_of(s), s.h_r, s.theta,
                           s.b, s.pl, s.h_d, s.gain);
    return hipGetLastError();
}

hipError_t launch_gain(const RisVecState& s, const RisVecParams&, hipStream_t st) {
    const int vec = (s.n_ris & 1) ? 1 : 2;
    switch (pick_group(s.n_ris, vec, kWave)) {
        case 8: return launch_gain_g<8>(s, st);
        case 16: return launch_gain_g<16>(s, st);
        case 32: return launch_gain_g<32>(s, st);
        default: return launch_gain_g<64>(s, st);
    }
}

template <int VP>
static hipError_t launch_step_vp(const RisVecState& s, const RisVecParams& p, const StepArgs& a,
                                 bool fused, hipStream_t st) {
    const long long threads = (long long)s.n_envs * VP;
    const unsigned grid = (unsigned)((threads + kBlock - 1) / kBlock);
    const Dims d = dims_of(s);
    if (!fused) {
        hipLaunchKernelGGL((k_step<VP>), dim3(grid), dim3(kBlock), 0, st, d, p, a);
        return hipGetLastError();
    }
    const bool even = (s.n_ris & 1) == 0;
    const int g = pick_group(s.n_ris, even ? 2 : 1, VP);
#define RISVEC_FUSED(GG)                                                                          \
    if (g == GG) {                                                                                \
        if (even) hipLaunchKernelGGL((k_step_fused<VP, GG, 2>), dim3(grid), dim3(kBlock), 0, st, d, p, a); \
        else hipLaunchKernelGGL((k_step_fused<VP, GG, 1>), dim3(grid), dim3(kBlock), 0, st, d, p, a);      \
        return hipGetLastError();                                                                 \
    }
    if constexpr (kWave / VP <= 8) { RISVEC_FUSED(8) }
    if constexpr (kWave / VP <= 16) { RISVEC_FUSED(16) }
    if constexpr (kWave / VP <= 32) { RISVEC_FUSED(32) }
    RISVEC_FUSED(64)
#undef RISVEC_FUSED
    return hipErrorInvalidValue;
}

hipError_t launch_step(const RisVecState& s, const RisVecParams& p, const float* action,
                       const int32_t* partner, const int32_t* n_groups, const int32_t* arrivals,
                       uint64_t seed, uint32_t counter, uint32_t flags, bool fused, hipStream_t st) {
    const StepArgs a = make_args(s, action, partner, n_groups, arrivals, seed, counter, flags);
    switch (pow2_ceil(s.n_veh)) {
        case 1: return launch_step_vp<1>(s, p, a, fused, st);
        case 2: return launch_step_vp<2>(s, p, a, fused, st);
        case 4: return launch_step_vp<4>(s, p, a, fused, st);
        case 8: return launch_step_vp<8>(s, p, a, fused, st);
        case 16: return launch_step_vp<16>(s, p, a, fused, st);
        case 32: return launch_step_vp<32>(s, p, a, fused, st);
        case 64: return launch_step_vp<64>(s, p, a, fused, st);
        default: return hipErrorInvalidValue;
    }
}

template <int VP>
static hipError_t launch_rate_vp(const RisVecState& s, const RisVecParams& p, const float* p_off,
                                 const int32_t* partner, const int32_t* n_groups, float* rate_out,
                                 hipStream_t st) {
    const long long threads = (long long)s.n_envs * VP;
    const unsigned grid = (unsigned)((threads + kBlock - 1) / kBlock);
    hipLaunchKernelGGL((k_data_rate<VP>), dim3(grid), dim3(kBlock), 0, st, dims_of(s), p, p_off, s.gain,
                       partner, n_groups, rate_out);
    return hipGetLastError();
}

hipError_t launch_data_rate(const RisVecState& s, const RisVecParams& p, const float* p_off,
                            const int32_t* partner, const int32_t* n_groups, float* rate_out,
                            hipStream_t st) {
    switch (pow2_ceil(s.n_veh)) {
        case 1: return launch_rate_vp<1>(s, p, p_off, partner, n_groups, rate_out, st);
        case 2: return launch_rate_vp<2>(s, p, p_off, partner, n_groups, rate_out, st);
        case 4: return launch_rate_vp<4>(s, p, p_off, partner, n_groups, rate_out, st);
        case 8: return launch_rate_vp<8>(s, p, p_off, partner, n_groups, rate_out, st);
        case 16: return launch_rate_vp<16>(s, p, p_off, partner, n_groups, rate_out, st);
        case 32: return launch_rate_vp<32>(s, p, p_off, partner, n_groups, rate_out, st);
        case 64: return launch_rate_vp<64>(s, p, p_off, partner, n_groups, rate_out, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace risvec
