// K5: one BCD (block-coordinate-descent) sweep over the discrete RIS phases.
//
// Reference: optimize_phase_shift ENV:208-220 with the objective of
// optimize_compute_objective_function ENV:222-231.  The reference objective sums
// the WHOLE [V,M] product (ENV:226 has no vehicle index), so it equals
//     Kc * | sum_m theta_m c_m |^2,   c_m = (sum_v h_r[v,m]) b[m],   Kc > 0,
// and the arg-max over the 2^b candidates for element m is that of
// | S - theta_m c_m + cand c_m |^2.  That turns the reference's O(M^2 2^b V^2) sweep
// into O(V M + M 2^b).  The sweep is a chain of M dependent discrete decisions, so
// it runs in float64 (inputs h_r, theta, b are the float32 tensors): a float32 sweep
// would flip near-tied decisions and drift away from the reference's theta.
//
// Mapping: a group of NC = 2^b lanes owns one env, lane k evaluates candidate k; the
// arg-max (first index wins ties, and the winner must score > 0: ENV:210-218) is a
// group butterfly.  c[] and theta[] of the envs of a block are staged in LDS.
#include "risvec_launch.hpp"

namespace risvec {

template <int NC>
__global__ void __launch_bounds__(kBlock)
k_bcd(Dims d, int epb, const float* __restrict__ h_r, float* __restrict__ theta,
      const float* __restrict__ b, int32_t* __restrict__ idx_out) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int M = d.M, V = d.V;
    double2* s_c = reinterpret_cast<double2*>(smem);                 // [epb][M]
    float2* s_th = reinterpret_cast<float2*>(s_c + (size_t)epb * M); // [epb][M]
    const int e_blk = blockIdx.x * epb;
    const int n_env = min(epb, d.E - e_blk);

    // phase 1: c[i][m] = (sum_v h_r[e,v,m]) * b[m]; stage theta
    for (int t = threadIdx.x; t < n_env * M; t += kBlock) {
        const int i = t / M, m = t % M;
        const long long e = e_blk + i;
        const float2* col = reinterpret_cast<const float2*>(h_r) + e * V * M + m;
        double sr = 0.0, si = 0.0;
        for (int v = 0; v < V; ++v) {
            const float2 h = col[(long long)v * M];
            sr += (double)h.x;
            si += (double)h.y;
        }
        const float2 bb = reinterpret_cast<const float2*>(b)[m];
        s_c[t] = make_double2(sr * bb.x - si * bb.y, sr * bb.y + si * bb.x);
        s_th[t] = reinterpret_cast<const float2*>(theta)[e * M + m];
    }
    __syncthreads();

    const int grp = threadIdx.x / NC, k = threadIdx.x % NC;
    const bool has_env = grp < n_env;          // uniform per group; NC | 64 so shuffles stay in-group
    const int gi = has_env ? grp : 0;
    const double2* c = s_c + (size_t)gi * M;
    const float2* th = s_th + (size_t)gi * M;
    const long long e = e_blk + gi;

    // phase 2: S = sum_m theta_m c_m
    double Sr = 0.0, Si = 0.0;
    if (has_env) {
        for (int m = k; m < M; m += NC) {
            const double2 cm = c[m];
            const double tr = th[m].x, ti = th[m].y;
            Sr += tr * cm.x - ti * cm.y;
            Si += tr * cm.y + ti * cm.x;
        }
    }
    Sr = group_sum<NC>(Sr);
    Si = group_sum<NC>(Si);

    // candidate k: exp(j 2 pi k / NC)   (ENV:169, 213)
    double cs, cc;
    sincospi(2.0 * (double)k / (double)NC, &cs, &cc);
    const int lane = threadIdx.x & (kWave - 1);
    const int base = lane - k;

    // phase 3: the sweep
    for (int m = 0; m < M; ++m) {
        const double2 cm = c[m];
        const double tr = th[m].x, ti = th[m].y;
        const double rr = Sr - (tr * cm.x - ti * cm.y);
        const double ri = Si - (tr * cm.y + ti * cm.x);
        const double pr = cc * cm.x - cs * cm.y, pi = cc * cm.y + cs * cm.x;   // cand * c_m
        const double zr = rr + pr, zi = ri + pi;
        double x = zr * zr + zi * zi;
        int kb = k;
#pragma unroll
        for (int o = NC / 2; o > 0; o >>= 1) {
            const double xo = __shfl_xor(x, o, kWave);
            const int ko = __shfl_xor(kb, o, kWave);
            if (xo > x || (xo == x && ko < kb)) { x = xo; kb = ko; }
        }
        const bool any = x > 0.0;                               // `best < x` from best = 0
        double nr = __shfl(cc, base + kb, kWave), ni = __shfl(cs, base + kb, kWave);
        if (!any) { nr = 0.0; ni = 0.0; kb = -1; }              // integer 0, ENV:211, 220
        Sr = rr + (nr * cm.x - ni * cm.y);
        Si = ri + (nr * cm.y + ni * cm.x);
        if (has_env && k == 0) {
            reinterpret_cast<float2*>(theta)[e * M + m] = make_float2((float)nr, (float)ni);
            if (idx_out) idx_out[e * M + m] = kb;
        }
    }
}

template <int NC>
static hipError_t launch_bcd_nc(const RisVecState& s, int32_t* idx_out, hipStream_t st) {
    const int M = s.n_ris;
    const size_t per_env = (size_t)M * (sizeof(double2) + sizeof(float2));
    int epb = kBlock / NC;
    const int cap = (int)((64u * 1024u) / per_env);
    if (epb > cap) epb = cap;
    if (epb < 1) return hipErrorInvalidValue;              // M > 2730: rejected by the API layer
    const unsigned grid = (unsigned)((s.n_envs + epb - 1) / epb);
    hipLaunchKernelGGL((k_bcd<NC>), dim3(grid), dim3(kBlock), epb * per_env, st, dims_of(s), epb,
                       s.h_r, s.theta, s.b, idx_out);
    return hipGetLastError();
}

hipError_t launch_bcd(const RisVecState& s, const RisVecParams&, int32_t* idx_out, hipStream_t st) {
    switch (s.control_bit) {
        case 0: return launch_bcd_nc<1>(s, idx_out, st);
        case 1: return launch_bcd_nc<2>(s, idx_out, st);
        case 2: return launch_bcd_nc<4>(s, idx_out, st);
        case 3: return launch_bcd_nc<8>(s, idx_out, st);
        case 4: return launch_bcd_nc<16>(s, idx_out, st);
        case 5: return launch_bcd_nc<32>(s, idx_out, st);
        case 6: return launch_bcd_nc<64>(s, idx_out, st);
        default: return hipErrorInvalidValue;
    }
}

// BCD + gains + step.  Round 1: two launches on the same stream (sweep, then the fused
// gain+step kernel); a single-launch variant with h_r resident in LDS is planned.
hipError_t launch_step_fused_bcd(const RisVecState& s, const RisVecParams& p, const float* action,
                                 const int32_t* partner, const int32_t* n_groups,
                                 const int32_t* arrivals, uint64_t seed, uint32_t counter,
                                 uint32_t flags, hipStream_t st) {
    hipError_t err = launch_bcd(s, p, nullptr, st);
    if (err != hipSuccess) return err;
    return launch_step(s, p, action, partner, n_groups, arrivals, seed, counter, flags, true, st);
}

}  // namespace risvec
