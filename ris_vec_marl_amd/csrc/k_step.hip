// The hot path, generic shapes: RIS cascaded gain (K3), step() (K4) and their fusion
// (K34) for ANY (V <= 64, M); compile-time shapes take the software-pipelined kernels of
// k_step_pipe.hip instead.
//
// Reference: Simulation-MARL-BCD/Environment.py (ENV): update_channel_gains "free"
// ENV:263-273, compute_data_rate ENV:331-372, step ENV:547-731.
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
#include <type_traits>

#include "risvec_step.hpp"

namespace risvec {

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
// RING: the env's replay transition is written from here too (StepArgs::ring; marl_train_bcd.py:1776-1799)
template <int VP, bool RING = false>
__global__ void __launch_bounds__(kBlock)
k_step(Dims d, RisVecParams P, StepArgs A) {
    RISVEC_ARGS_IN_ONE_TRIP("s"(d.E), "s"(d.V), "s"(A.flags), "s"(A.action), "s"(A.data_buf), "s"(A.partner), "s"(A.n_groups),
                            "s"(A.mec_q), "s"(A.gain), "s"(A.pl), "s"(A.arrivals));
    RISVEC_ARGS_IN_ONE_TRIP(RISVEC_STEP_PARAMS(P));
    const long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
    const int e = (int)(t / VP), v = (int)(t % VP);
    const bool active = e < d.E && v < d.V;
    const StepIn in = load_step_in(d, A, e, v, active);
    const float g = active ? A.gain[(long long)e * d.V + v] : 0.f;
    if constexpr (RING) {
        const RingIn<VP> rin = load_ring_in<VP>(d, A, e, v, active);
        step_core<VP, false, true, RingIn<VP>>(d, P, A, e, v, active, g, in, nullptr, &rin);
    } else {
        step_core<VP, false, true>(d, P, A, e, v, active, g, in);
    }
}

// K4 over T steps: the driver's own cadence (marl_train_bcd.py:1304-1611: step() every step, the channel gains only every
// K_STEPS_FOR_RIS_OPTIMIZATION = 100 steps) in ONE launch -- n_steps consecutive step() calls on the cached gains, for
// any shape.  A lane owns one (env, vehicle) for the whole launch; the env's queues stay in registers, every step's
// records go to their slice of the trajectory buffers, the env's own tensors receive the last step's outputs:
// bit-identical to n_steps launches of k_step.
template <int VP>
__global__ void __launch_bounds__(kBlock)
k_step_multi(Dims d, RisVecParams P, StepArgs A, int n_steps, RisVecTraj TJ) {
    RISVEC_ARGS_IN_ONE_TRIP("s"(d.E), "s"(d.V), "s"(A.flags), "s"(A.action), "s"(A.data_buf), "s"(A.partner), "s"(A.n_groups),
                            "s"(A.mec_q), "s"(A.gain), "s"(A.pl), "s"(A.arrivals), "s"(n_steps));
    const long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
    const int e = (int)(t / VP), v = (int)(t % VP);
    const bool active = e < d.E && v < d.V;
    const StepIn in = load_step_in(d, A, e, v, active);
    const float g = active ? A.gain[(long long)e * d.V + v] : 0.f;
    multi_step_loop<VP>(d, P, A, TJ, e, v, active, g, in, n_steps);
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

// K34, generic: fused gain + step.  Each wave owns 64/VP consecutive envs.
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

    const int e_mine = e0 + lane / VP, v_mine = lane % VP;
    const bool active = e_mine < d.E && v_mine < V;
    const StepIn in = load_step_in(d, A, e_mine, v_mine, active);   // issued ahead of the cascade

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

    float g = 0.f;
    if (active) {
        const long long idx = (long long)e_mine * V + v_mine;
        g = gain_from_img(s_img[wave][lane], in.pl, A.h_d, idx);
        A.gain[idx] = g;
    }
    step_core<VP>(d, P, A, e_mine, v_mine, active, g, in);
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------

// "Steering" form of the fused kernel (RISVEC_STEP_STEER).  compute_parms makes every row of h_r a
// geometric sequence, h_r[e,v,m] = z^m with z = exp(-j pi angle_v) (ENV:249-253), so the cascade
// sum_m theta_m b_m h_r[e,v,m] is the polynomial sum_m w_m z^m, w = theta * b.  Instead of streaming the
// 8M-byte row from HBM, a lane reads its 16-byte z (float64) and evaluates the polynomial by Horner's
// rule in float64 -- four interleaved chains in z^4, so the dependent chain is M/4 long -- against the
// env's w row staged once per wavefront in LDS.  Accumulated error ~M * 2^-53: better than the float32
// sum over the stored float32 row.  Per env-step the kernel moves 16V + 8M + 64V + 68 bytes instead of
// 8VM + 8M + 64V + 68 (1 220 vs 5 188 at V = 8, M = 64).
// WIDE: the staged w row is widened to float64 once (saves two conversions per Horner step) when 4
// wavefronts' worth of it still leaves room for several blocks per CU; long rows stay float32 in LDS.
template <int VP, bool WIDE>
__global__ void __launch_bounds__(kBlock)
k_step_steer(Dims d, RisVecParams P, StepArgs A, const double* __restrict__ z_r) {
    constexpr int EPW = kWave / VP;                        // envs per wave
    using W2 = typename std::conditional<WIDE, double2, float2>::type;
    extern __shared__ double2 s_wrow_raw[];                // [waves][EPW][M + 1]   w = theta * b (padded rows)
    W2* s_wrow = reinterpret_cast<W2*>(s_wrow_raw);
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int V = d.V, M = d.M, MS = M + 1;
    const int e0 = (blockIdx.x * (blockDim.x / kWave) + wave) * EPW;
    W2* sw = s_wrow + (long long)wave * EPW * MS;
    const int e_mine = e0 + lane / VP, v_mine = lane % VP;
    const bool active = e_mine < d.E && v_mine < V;
    const StepIn in = load_step_in(d, A, e_mine, v_mine, active);
    double2 z = make_double2(1.0, 0.0);
    if (active) z = reinterpret_cast<const double2*>(z_r)[(long long)e_mine * V + v_mine];
    for (int idx = lane; idx < EPW * M; idx += kWave) {   // coalesced theta rows of the wave's envs
        const int i = idx / M, m = idx - i * M;
        const int e = e0 + i;
        float2 w = make_float2(0.f, 0.f);
        if (e < d.E) {
            const float2 t = *reinterpret_cast<const float2*>(A.theta + ((long long)e * M + m) * 2);
            w = cmul(t, *reinterpret_cast<const float2*>(A.b + m * 2));
        }
        if constexpr (WIDE) sw[i * MS + m] = make_double2((double)w.x, (double)w.y);
        else sw[i * MS + m] = w;
    }
    __syncthreads();
    // z^2, z^4 and four Horner chains over m = 4k + r, highest power first
    const double z2r = z.x * z.x - z.y * z.y, z2i = 2.0 * z.x * z.y;
    const double z4r = z2r * z2r - z2i * z2i, z4i = 2.0 * z2r * z2i;
    const W2* wr = sw + (lane / VP) * MS;
    double ar[4] = {0.0, 0.0, 0.0, 0.0}, ai[4] = {0.0, 0.0, 0.0, 0.0};
    const int M4 = (M + 3) / 4;
    for (int k = M4 - 1; k >= 0; --k) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = 4 * k + r;
            double wx = 0.0, wy = 0.0;
            if (m < M) { const W2 w = wr[m]; wx = (double)w.x; wy = (double)w.y; }
            const double nr = ar[r] * z4r - ai[r] * z4i + wx;
            const double ni = ar[r] * z4i + ai[r] * z4r + wy;
            ar[r] = nr; ai[r] = ni;
        }
    }
    // sum_r z^r acc_r = acc0 + z (acc1 + z (acc2 + z acc3))
    double sr = ar[3], si = ai[3];
#pragma unroll
    for (int r = 2; r >= 0; --r) {
        const double nr = sr * z.x - si * z.y + ar[r], ni = sr * z.y + si * z.x + ai[r];
        sr = nr; si = ni;
    }
    float g = 0.f;
    if (active) {
        const long long idx = (long long)e_mine * V + v_mine;
        g = gain_from_img(make_float2((float)sr, (float)si), in.pl, A.h_d, idx);
        A.gain[idx] = g;
    }
    step_core<VP>(d, P, A, e_mine, v_mine, active, g, in);
}

template <int VP>
static hipError_t launch_steer_vp(const RisVecState& s, const RisVecParams& p, const StepArgs& a, hipStream_t st) {
    constexpr int EPW = kWave / VP;
    const size_t row = (size_t)EPW * (s.n_ris + 1);
    const bool wide = row * sizeof(double2) * (kBlock / kWave) <= 40 * 1024;     // >= 4 blocks per CU either way
    const size_t per_wave = row * (wide ? sizeof(double2) : sizeof(float2));
    int waves = (int)((64 * 1024) / per_wave);             // as many wavefronts per block as 64 KB of LDS hold
    if (waves < 1) return hipErrorNotSupported;            // theta rows do not fit: use the streaming kernel
    if (waves > kBlock / kWave) waves = kBlock / kWave;
    const long long n_waves = ((long long)s.n_envs + EPW - 1) / EPW;
    const dim3 grid((unsigned)((n_waves + waves - 1) / waves)), block(waves * kWave);
    if (wide) hipLaunchKernelGGL((k_step_steer<VP, true>), grid, block, per_wave * waves, st, dims_of(s), p, a, s.z_r);
    else hipLaunchKernelGGL((k_step_steer<VP, false>), grid, block, per_wave * waves, st, dims_of(s), p, a, s.z_r);
    note_kernel("k_step_steer<%d,%s>", VP, wide ? "wide" : "narrow");
    return hipGetLastError();
}

hipError_t launch_step_steer(const RisVecState& s, const RisVecParams& p, const StepArgs& a, hipStream_t st) {
    switch (pow2_ceil(s.n_veh)) {
        case 1: return launch_steer_vp<1>(s, p, a, st);
        case 2: return launch_steer_vp<2>(s, p, a, st);
        case 4: return launch_steer_vp<4>(s, p, a, st);
        case 8: return launch_steer_vp<8>(s, p, a, st);
        case 16: return launch_steer_vp<16>(s, p, a, st);
        case 32: return launch_steer_vp<32>(s, p, a, st);
        case 64: return launch_steer_vp<64>(s, p, a, st);
        default: return hipErrorInvalidValue;
    }
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
    {
        const hipError_t err = launch_gain_pipe(s, st);     // compile-time shapes: pipelined form
        if (err != hipErrorNotSupported) return err;
    }
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
        if (a.ring.state_memory) {
            if constexpr (VP == 4 || VP == 8 || VP == 16) {
                hipLaunchKernelGGL((k_step<VP, true>), dim3(grid), dim3(kBlock), 0, st, d, p, a);
                note_kernel("k_step<%d,RING>", VP);
                return hipGetLastError();
            } else {
                return hipErrorNotSupported;
            }
        }
        hipLaunchKernelGGL((k_step<VP>), dim3(grid), dim3(kBlock), 0, st, d, p, a);
        note_kernel("k_step<%d>", VP);
        return hipGetLastError();
    }
    if (a.ring.state_memory) return hipErrorNotSupported;      // fused + ring: the software pipeline's ring form only
    const bool even = (s.n_ris & 1) == 0;
    const int g = pick_group(s.n_ris, even ? 2 : 1, VP);
#define RISVEC_FUSED(GG)                                                                          \
    if (g == GG) {                                                                                \
        if (even) hipLaunchKernelGGL((k_step_fused<VP, GG, 2>), dim3(grid), dim3(kBlock), 0, st, d, p, a); \
        else hipLaunchKernelGGL((k_step_fused<VP, GG, 1>), dim3(grid), dim3(kBlock), 0, st, d, p, a);      \
        note_kernel("k_step_fused<%d,%d,%d>", VP, GG, even ? 2 : 1);                                \
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
                       uint64_t seed, uint32_t counter, uint32_t flags, bool fused, hipStream_t st, const StepRing* ring) {
    StepArgs a = make_step_args(s, action, partner, n_groups, arrivals, seed, counter,
                                flags & ~(uint32_t)(RISVEC_STEP_STEER | RISVEC_STEP_THETA_BY_INDEX));
    if (ring) {
        // the transition store rides in the step kernel: the software pipeline's ring form (compile-time shapes) or the
        // cached-gain k_step<VP, RING>; no other member of the family carries it
        a.ring = *ring;
        if (fused) return launch_step_fused_pipe_ring(s, p, a, st);
    }
    if (fused && (flags & RISVEC_STEP_THETA_BY_INDEX)) {
        // theta is kept by index: only the latency-shaped family has that form (the API checked the shape)
        a.theta_k = s.theta_idx;
        a.theta_k_stride = theta_idx_stride(s.n_ris);
        return launch_step_fused_lat(s, p, a, st);
    }
    if (fused && (flags & RISVEC_STEP_STEER)) {
        const hipError_t err = launch_step_steer(s, p, a, st);
        if (err != hipErrorNotSupported) return err;       // theta rows too long for LDS: stream h_r as usual
    }
    if (fused) {                                                // small batch: latency-shaped single-group kernel
        const hipError_t err = launch_step_fused_lat(s, p, a, st);
        if (err != hipErrorNotSupported) return err;
    }
    if (fused) {
        const hipError_t err = launch_step_fused_pipe(s, p, a, st);
        if (err != hipErrorNotSupported) return err;
    }
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
static hipError_t launch_step_multi_vp(const RisVecState& s, const RisVecParams& p, const StepArgs& a, int n_steps,
                                       const RisVecTraj& tj, hipStream_t st) {
    const long long threads = (long long)s.n_envs * VP;
    const unsigned grid = (unsigned)((threads + kBlock - 1) / kBlock);
    hipLaunchKernelGGL((k_step_multi<VP>), dim3(grid), dim3(kBlock), 0, st, dims_of(s), p, a, n_steps, tj);
    note_kernel("k_step_multi<%d>", VP);
    return hipGetLastError();
}

hipError_t launch_step_multi(const RisVecState& s, const RisVecParams& p, const StepArgs& a, int n_steps,
                             const RisVecTraj* traj, hipStream_t st) {
    const RisVecTraj tj = traj ? *traj : RisVecTraj{nullptr, nullptr, nullptr};
    switch (pow2_ceil(s.n_veh)) {
        case 1: return launch_step_multi_vp<1>(s, p, a, n_steps, tj, st);
        case 2: return launch_step_multi_vp<2>(s, p, a, n_steps, tj, st);
        case 4: return launch_step_multi_vp<4>(s, p, a, n_steps, tj, st);
        case 8: return launch_step_multi_vp<8>(s, p, a, n_steps, tj, st);
        case 16: return launch_step_multi_vp<16>(s, p, a, n_steps, tj, st);
        case 32: return launch_step_multi_vp<32>(s, p, a, n_steps, tj, st);
        case 64: return launch_step_multi_vp<64>(s, p, a, n_steps, tj, st);
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
