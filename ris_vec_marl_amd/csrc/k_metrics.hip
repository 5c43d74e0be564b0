// f4: the per-episode metrics sink -- what marl_train_bcd.py (TRAIN) does in Python floats
// around every env.step: sum the last_* scalars (TRAIN:1626-1662), the clipped per-user rewards
// (TRAIN:1714, 1769) and the equivalent powers (TRAIN:1717-1753), and at the end of the episode turn
// them into the TensorBoard scalars (TRAIN:1824-1865, 1939-1941, 1927-2048) -- for E envs at once.
//   * k_episode_accumulate: one lane per env and column segment over a column-major accumulator:
//     everything coalesced, float64 sums in the driver's own order (step after step), so an env's
//     episode sums are those of the reference;
//   * k_episode_summary: one lane per env computes its RISVEC_EP_COLS scalars, a workgroup folds 256
//     envs (DPP butterflies on doubles, then the four waves through LDS) into one partial row;
//   * k_episode_fold: one workgroup folds the partial rows in index order -> mean / min / max over
//     the envs, deterministic whatever the grid was.
// HBM-bound byte work (float64 read-modify-write of E x (17+V) accumulators per step).
#include "risvec_launch.hpp"
#include "risvec_step.hpp"

namespace risvec {
namespace {

constexpr int kFixed = RISVEC_EP_FIXED;
constexpr int kCols = RISVEC_EP_COLS;
constexpr int kBestCol = 16;
constexpr double kInf = __builtin_huge_val();

// acc is column-major, [17+V][E]: a lane owns one env, so every accumulator column is read and written
// as 512 contiguous bytes per wavefront and a lane keeps all its read-modify-writes in flight at once.
__global__ void __launch_bounds__(kBlock)
k_episode_clear(int E, int A, double* __restrict__ acc) {
    const long long g = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (g >= (long long)E * A) return;
    acc[g] = g / E == kBestCol ? -kInf : 0.0;
}

// grid.y picks the segment: 0 = metrics slots 0..7, 1 = slots 8..13 + the best global reward,
// 2 = the two power sums, 3 = the V clipped per-user rewards.  One wavefront per workgroup so that
// E = 32 768 still gives 2 048 of them.
__global__ void __launch_bounds__(kWave)
k_episode_accumulate(int E, int V, const float* __restrict__ metrics, const float* __restrict__ reward,
                     const float* __restrict__ power_w, float user_clip, double* __restrict__ acc) {
    RISVEC_ARGS_IN_ONE_TRIP("s"(E), "s"(V), "s"(metrics), "s"(reward), "s"(power_w), "s"(acc));
    const int e = blockIdx.x * kWave + threadIdx.x;
    if (e >= E) return;
    const size_t S = (size_t)E;
    double* col = acc + e;
    const float4* m4 = reinterpret_cast<const float4*>(metrics + (size_t)e * RISVEC_METRICS);
    if (blockIdx.y == 0) {
        const float4 a = m4[0], b = m4[1];
        const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        double old[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) old[k] = col[k * S];
#pragma unroll
        for (int k = 0; k < 8; ++k) col[k * S] = old[k] + (double)x[k];
    } else if (blockIdx.y == 1) {
        const float4 a = m4[2], b = m4[3];
        const float g = metrics[(size_t)e * RISVEC_METRICS + RISVEC_METRIC_GLOBAL_REWARD];
        const float x[6] = {a.x, a.y, a.z, a.w, b.x, b.y};
        double old[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) old[k] = col[(8 + k) * S];
        const double best = col[kBestCol * S];
#pragma unroll
        for (int k = 0; k < 6; ++k) col[(8 + k) * S] = old[k] + (double)x[k];
        if ((double)g > best) col[kBestCol * S] = (double)g;      // ep_env_best (TRAIN:1613-1622)
    } else if (blockIdx.y == 2) {
        if (!power_w) return;
        const float* row = power_w + (size_t)e * 2 * V;
        const double o0 = col[14 * S], o1 = col[15 * S];
        double s0 = 0.0, s1 = 0.0;                                // np.sum of the row, left to right
        for (int v = 0; v < V; ++v) { s0 += (double)row[v]; s1 += (double)row[V + v]; }
        col[14 * S] = o0 + s0;                                    // offload (TRAIN:1752)
        col[15 * S] = o1 + s1;                                    // local   (TRAIN:1753)
    } else {
        const float* row = reward + (size_t)e * V;
        for (int v0 = 0; v0 < V; v0 += 8) {                       // eight read-modify-writes in flight
            double old[8];
            float r[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (v0 + k < V) { old[k] = col[(kFixed + v0 + k) * S]; r[k] = row[v0 + k]; }
#pragma unroll
            for (int k = 0; k < 8; ++k)                           // np.clip(per_user_reward, -5, 5) (TRAIN:1714)
                if (v0 + k < V) col[(kFixed + v0 + k) * S] = old[k] + (double)fminf(fmaxf(r[k], -user_clip), user_clip);
        }
    }
}

template <int OP> __device__ __forceinline__ double combine(double a, double b) {
    if constexpr (OP == 0) return a + b;
    else if constexpr (OP == 1) return fmin(a, b);
    else return fmax(a, b);
}
template <int OP> __device__ __forceinline__ double wave_fold(double x) {
    x = combine<OP>(x, xchg<1>(x)); x = combine<OP>(x, xchg<2>(x)); x = combine<OP>(x, xchg<4>(x));
    x = combine<OP>(x, xchg<8>(x)); x = combine<OP>(x, xchg<16>(x)); x = combine<OP>(x, xchg<32>(x));
    return x;
}

__global__ void __launch_bounds__(kBlock)
k_episode_summary(int E, int V, int n_steps, const double* __restrict__ acc, const float* __restrict__ metrics,
                  double* __restrict__ per_env, double* __restrict__ partial) {
    __shared__ double lds[kBlock / kWave][3][kCols];
    const int e = blockIdx.x * kBlock + threadIdx.x;
    const bool live = e < E;
    double c[kCols];
#pragma unroll
    for (int k = 0; k < kCols; ++k) c[k] = 0.0;
    if (live) {
        const size_t S = (size_t)E;
        const double* col = acc + e;
        auto row = [&](int a) { return col[a * S]; };
        const double n = (double)n_steps;
#pragma unroll
        for (int k = 0; k < 14; ++k) c[k] = row(k) / n;                         // TRAIN:1838, 1850-1865
        c[RISVEC_EP_OFF_KBIT] = row(1);                                          // sums, TRAIN:2047-2048
        c[RISVEC_EP_LOCAL_KBIT] = row(2);
        c[RISVEC_EP_MEC_CYCLES] = (double)metrics[(long long)e * RISVEC_METRICS + RISVEC_METRIC_MEC_QUEUE];
        const double p_off = row(14), p_loc = row(15);
        c[RISVEC_EP_POWER_OFFLOAD] = p_off / n;                                  // np.mean(Power_offload)
        c[RISVEC_EP_POWER_LOCAL] = p_loc / n;
        c[RISVEC_EP_POWER_TOTAL] = (p_off + p_loc) / n;                          // np.mean(Power), TRAIN:1719
        double s = 0.0, s2 = 0.0, mn = kInf;
        for (int v = 0; v < V; ++v) {                                            // record_reward_[:, ep] / n (TRAIN:1824)
            const double x = row(kFixed + v) / n;
            s += x; s2 += x * x; mn = fmin(mn, x);
        }
        const double mean = s / V;
        double var = 0.0;
        for (int v = 0; v < V; ++v) { const double d = row(kFixed + v) / n - mean; var += d * d; }
        c[RISVEC_EP_MIN_USER] = mn;
        c[RISVEC_EP_VAR_USER] = var / V;                                         // np.var (TRAIN:1940)
        c[RISVEC_EP_JAIN] = s * s / ((double)V * s2 + 1e-12);                    // _jain_index (TRAIN:112-119)
        c[RISVEC_EP_BEST_GLOBAL] = row(kBestCol);
        if (per_env) {
#pragma unroll
            for (int k = 0; k < kCols; ++k) per_env[(long long)e * kCols + k] = c[k];
        }
    }
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
#pragma unroll
    for (int k = 0; k < kCols; ++k) {
        const double sum = wave_fold<0>(c[k]);
        const double lo = wave_fold<1>(live ? c[k] : kInf);
        const double hi = wave_fold<2>(live ? c[k] : -kInf);
        if (lane == 0) { lds[wave][0][k] = sum; lds[wave][1][k] = lo; lds[wave][2][k] = hi; }
    }
    __syncthreads();
    if (threadIdx.x < 3 * kCols) {
        const int op = threadIdx.x / kCols, k = threadIdx.x % kCols;
        double x = lds[0][op][k];
        for (int w = 1; w < kBlock / kWave; ++w)
            x = op == 0 ? x + lds[w][op][k] : (op == 1 ? fmin(x, lds[w][op][k]) : fmax(x, lds[w][op][k]));
        partial[((long long)blockIdx.x * 3 + op) * kCols + k] = x;
    }
}

__global__ void __launch_bounds__(kWave)
k_episode_fold(int E, int rows, const double* __restrict__ partial, double* __restrict__ summary) {
    const int t = threadIdx.x;
    if (t >= 3 * kCols) return;
    const int op = t / kCols;
    double x = partial[t];
    for (int r0 = 1; r0 < rows; r0 += 16) {                      // 16 loads in flight, folded in index order
        double y[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) y[k] = r0 + k < rows ? partial[(long long)(r0 + k) * 3 * kCols + t] : 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (r0 + k < rows) x = op == 0 ? x + y[k] : (op == 1 ? fmin(x, y[k]) : fmax(x, y[k]));
    }
    summary[t] = op == 0 ? x / (double)E : x;
}

}  // namespace

int episode_partial_rows(int E) { return (E + kBlock - 1) / kBlock; }

hipError_t launch_episode_clear(int E, int V, double* acc, hipStream_t st) {
    const long long n = (long long)E * (kFixed + V);
    hipLaunchKernelGGL(k_episode_clear, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, E, kFixed + V, acc);
    return hipGetLastError();
}

hipError_t launch_episode_accumulate(int E, int V, const float* metrics, const float* reward, const float* power_w,
                                     float user_clip, double* acc, hipStream_t st) {
    hipLaunchKernelGGL(k_episode_accumulate, dim3((E + kWave - 1) / kWave, 4), dim3(kWave), 0, st, E, V, metrics, reward,
                       power_w, user_clip, acc);
    return hipGetLastError();
}

hipError_t launch_episode_summary(int E, int V, int n_steps, const double* acc, const float* metrics, double* per_env,
                                  double* partial, double* summary, hipStream_t st) {
    const int rows = episode_partial_rows(E);
    hipLaunchKernelGGL(k_episode_summary, dim3(rows), dim3(kBlock), 0, st, E, V, n_steps, acc, metrics, per_env, partial);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return err;
    static_assert(3 * kCols <= kWave, "one wavefront folds every (op, column)");
    hipLaunchKernelGGL(k_episode_fold, dim3(1), dim3(kWave), 0, st, E, rows, partial, summary);
    return hipGetLastError();
}

}  // namespace risvec
