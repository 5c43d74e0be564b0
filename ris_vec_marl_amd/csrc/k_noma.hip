// f2: the NOMA grouping stage of the reference driver, marl_train_bcd.py (TRAIN) --
// the work done for every env right before env.step(): |delta g_dB| feasibility mask
// (TRAIN:128-156, 842-855), score matrix (TRAIN:164-194), quantile-gated max-weight matching
// (TRAIN:326-398), greedy completion (TRAIN:276-324), mask relaxation (TRAIN:260-275), pair QoS
// check (TRAIN:858-880) and the control logic with the freeze-in-episode safeties
// (TRAIN:1401-1562, 1618-1623).
//
// Work decomposition: ONE WAVEFRONT PER ENV (workgroup = 64 lanes), grid-stride over envs.  The
// N x N matrices (N <= 16) live in LDS with the entries dealt round-robin to the lanes; every
// decision the reference takes by comparing float64 numbers is taken here by comparing float64
// numbers formed in the same association order (fp contraction is off for the whole file: the
// matcher's exact ties are decided by last-bit rounding).
//   * quantiles: no sort -- each lane ranks its own entries against all (LDS broadcast reads) and
//     the two order statistics NumPy's linear method interpolates are picked by rank;
//   * matching: the reference's memoised recursion on the lowest unused user, evaluated bottom-up
//     over bitmasks in popcount layers (a mask only needs masks with 1 or 2 more bits), after
//     dropping users without any admissible edge (they pass the value through unchanged); table
//     in LDS up to 2^KL entries, in a caller-provided HBM slot beyond;
//   * completion: repeated wave-wide arg-max over the still-free admissible edges, which is what
//     the reference's sorted greedy scan selects.
// This is integer / branchy float64 work of a few KB per env; it is latency-bound, not HBM-bound.
#include "risvec_launch.hpp"

#pragma clang fp contract(off)

namespace risvec {
namespace {

constexpr int kNV = RISVEC_NOMA_MAX_VEH;
constexpr int kNN = kNV * kNV;
constexpr int kEPL = kNN / kWave;                   // matrix entries per lane (4)
constexpr uint32_t kSiteUnstick = 7;                // Philox site of the TRAIN:1539 draw
constexpr double kInf = __builtin_huge_val();

struct NomaArgs {
    RisVecNomaState ns;
    RisVecNomaParams P;
    const float* gain;
    const double* gdb12;
    const float* p01;
    int use_mask;
    double q_back;
    int K_back;
    const double* tau_back;
    const float* prev_global;
    int prev_stride;
    int i_step;
    const float* u_unstick;
    uint64_t seed;
    uint32_t counter;
    int32_t* partner_out;
    int32_t* n_groups_out;
    int32_t* info_out;
};

__device__ __forceinline__ double wave_max(double x) {
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) x = fmax(x, __shfl_xor(x, o, kWave));
    return x;
}
__device__ __forceinline__ int wave_sum(int x) {
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) x += __shfl_xor(x, o, kWave);
    return x;
}
__device__ __forceinline__ bool finite(double x) { return fabs(x) < kInf; }

// Rank bookkeeping of one lane's entries within the multiset {val[k] : ok[k]}.
struct Ranks {
    double x[kEPL];
    int less[kEPL], leq[kEPL];
    bool ok[kEPL];
    int cnt;
};

__device__ __forceinline__ void rank_entries(const double* val, const uint8_t* ok, int n, int lane, Ranks& R) {
    int mine = 0;
#pragma unroll
    for (int t = 0; t < kEPL; ++t) {
        const int idx = lane + t * kWave;
        R.ok[t] = idx < n && ok[idx];
        R.x[t] = R.ok[t] ? val[idx] : 0.0;
        R.less[t] = R.leq[t] = 0;
        mine += R.ok[t] ? 1 : 0;
    }
    for (int k = 0; k < n; ++k) {                     // broadcast reads: every lane the same address
        if (!ok[k]) continue;
        const double y = val[k];
#pragma unroll
        for (int t = 0; t < kEPL; ++t) {
            R.less[t] += y < R.x[t] ? 1 : 0;
            R.leq[t] += y <= R.x[t] ? 1 : 0;
        }
    }
    R.cnt = wave_sum(mine);
}

// k-th smallest (0-based) of the ranked multiset.
__device__ __forceinline__ double order_stat(const Ranks& R, int k) {
    double v = -kInf;
#pragma unroll
    for (int t = 0; t < kEPL; ++t)
        if (R.ok[t] && R.less[t] <= k && k < R.leq[t]) v = R.x[t];
    return wave_max(v);
}

// np.quantile(values, q), method 'linear': virtual index (n-1) q, neighbours floor / floor+1 (both
// the last element from n-1 up), two-sided lerp (a + d t below t = 0.5, b - d (1-t) from there).
__device__ __forceinline__ double quantile_linear(const Ranks& R, double q) {
    const int n = R.cnt;
    const double vi = (double)(n - 1) * q;
    int lo = (int)floor(vi), hi;
    double t = vi - floor(vi);
    hi = lo + 1;
    if (vi >= (double)(n - 1)) { lo = hi = n - 1; t = 0.0; }
    if (vi < 0.0) { lo = hi = 0; t = 0.0; }
    const double a = order_stat(R, lo), b = order_stat(R, hi);
    const double d = b - a;
    double r = a + d * t;
    if (t >= 0.5) r = b - d * (1.0 - t);
    return r;
}

// Value and choice of the matching recurrence at `mask` (TRAIN:360-391): lowest unused user x stays
// single (arg -1) or takes partner j (arg j); arg -2 = no option (value 0, nothing below it).
__device__ __forceinline__ double best_at(const double* T, const double* w, int K, bool singles, int mask,
                                          int& arg) {
    const int x = __ffs(~mask) - 1;
    double best = -kInf;
    arg = -2;
    if (singles) {
        const double w1 = T[mask | (1 << x)];
        if (w1 > best) { best = w1; arg = -1; }
    }
    for (int j = x + 1; j < K; ++j) {
        if ((mask >> j) & 1) continue;
        const double we = w[x * kNV + j];
        if (!finite(we)) continue;
        const double w2 = T[mask | (1 << x) | (1 << j)];
        if (finite(w2) && we + w2 > best) { best = we + w2; arg = j; }
    }
    return best;
}

template <int KL>
__global__ void __launch_bounds__(kWave)
k_noma_group(NomaArgs A) {
    __shared__ double s_S[kNN], s_W[kNN], s_w[kNN], s_g[kNV], s_lin[kNV], s_p[kNV], s_dp[1 << KL];
    __shared__ float s_hist[kNN];
    __shared__ uint8_t s_feas[kNN], s_ok[kNN], s_qos[kNN];
    __shared__ int s_part[kNV];
    const int lane = threadIdx.x;
    const int N = A.ns.n_veh, NN = N * N, E = A.ns.n_envs;
    const RisVecNomaParams& P = A.P;
    const bool singles = P.mwm_allow_singles != 0;

    for (int env = blockIdx.x; env < E; env += gridDim.x) {
        __syncthreads();                               // LDS reuse across envs
        // ---- ep_env_best / last_env_global of the previous step (TRAIN:1618-1623) ----------
        int flags = A.ns.flags[env];
        double last = A.ns.last_global[env], best = A.ns.best_global[env];
        if (A.prev_global) {
            const double g = (double)A.prev_global[(long long)env * A.prev_stride];
            if (!(flags & RISVEC_NOMA_HAS_LAST)) best = g;
            else if (g > best) best = g;
            last = g;
            flags |= RISVEC_NOMA_HAS_LAST;
        }
        // ---- history decay, float32 in place (TRAIN:1406) ------------------------------------
        float* hist = A.ns.hist + (long long)env * NN;
        for (int idx = lane; idx < NN; idx += kWave) s_hist[idx] = hist[idx] * P.pair_hist_decay;
        // ---- freeze-in-episode with its three safeties (TRAIN:1527-1540) -----------------------
        const bool frozen = P.freeze_group_in_episode && (flags & RISVEC_NOMA_HAS_GROUPS);
        bool need_repair = false;
        if (frozen) {
            if (P.freeze_recalc_every > 0 && A.i_step % P.freeze_recalc_every == 0) need_repair = true;
            if (!need_repair && (flags & RISVEC_NOMA_HAS_LAST) && !(flags & RISVEC_NOMA_UNSTICK_USED)) {
                if (last < best * (1.0 - P.freeze_reward_drop_ratio)) {
                    need_repair = true;
                    flags |= RISVEC_NOMA_UNSTICK_USED;
                }
            }
            if (!need_repair && P.freeze_unstick_prob > 0.0) {
                const double u = A.u_unstick
                    ? (double)A.u_unstick[env]
                    : (double)u01(philox4x32_10((uint32_t)(A.ns.env_offset + env), 0u, A.counter, kSiteUnstick, A.seed).x);
                if (u < P.freeze_unstick_prob) need_repair = true;
            }
        }
        int rounds = 0, npairs = 0, K_last = 0;
        const bool recompute = !(frozen && !need_repair);
        if (!recompute) {                              // reuse episode_groups (TRAIN:1542-1547)
            if (lane < N) s_part[lane] = A.ns.partner[(long long)env * N + lane];
            __syncthreads();
            for (int v = 0; v < N; ++v) npairs += (s_part[v] >= 0 && s_part[v] < 65536) ? 1 : 0;
        } else {
            // ================= solve (TRAIN:1419-1524) ==========================================
            if (lane < N) {
                const double g = (double)A.gain[(long long)env * N + lane];
                s_lin[lane] = g;
                s_g[lane] = A.gdb12 ? A.gdb12[(long long)env * N + lane] : 10.0 * log10(fmax(g, 1e-12));
                s_p[lane] = A.p01 ? (double)A.p01[(long long)env * N + lane] : 0.0;
            }
            for (int idx = lane; idx < NN; idx += kWave) {
                const int i = idx / N, j = idx % N;
                s_feas[idx] = (P.mask_enable && A.use_mask) ? A.ns.mask[(long long)env * NN + idx] : (i != j);
            }
            __syncthreads();
            if (P.qos_enable) {                        // TRAIN:1426-1441 + 858-880
                for (int idx = lane; idx < NN; idx += kWave) {
                    const int i = idx / N, j = idx % N;
                    bool okq = false;
                    if (i != j) {
                        const double pi = s_p[i] * P.P_max, pj = s_p[j] * P.P_max;
                        const double gi = s_lin[i], gj = s_lin[j];
                        const bool inear = gi >= gj;
                        const double gn = inear ? gi : gj, gf = inear ? gj : gi;
                        const double pn = inear ? pi : pj, pf = inear ? pj : pi;
                        const double sf = (pf * gf) / (pn * gf + P.noise_power + 1e-12);
                        const double sn = (pn * gn) / (P.noise_power + 1e-12);
                        okq = log2(1.0 + fmax(0.0, sf)) >= P.qos_R_min && log2(1.0 + fmax(0.0, sn)) >= P.qos_R_min;
                    }
                    s_qos[idx] = okq;
                }
            }
            const int target = max(1, P.min_pair_target);
            double accept_q = P.mwm_accept_quantile;
            int K_back = A.K_back;
            double tau_b = A.tau_back[env];
            unsigned busy = 0;                         // wave-uniform: users already paired
            unsigned long long mate = 0;               // 4 bits per user, valid where busy
            while (true) {
                // ---- score matrix (TRAIN:164-194) -----------------------------------------------
                bool any_ok = false;
                for (int idx = lane; idx < NN; idx += kWave) {
                    const int i = idx / N, j = idx % N;
                    const bool abs_ok = s_g[i] >= P.abs_gain_min_db || s_g[j] >= P.abs_gain_min_db;
                    any_ok = any_ok || (s_feas[idx] && abs_ok);
                }
                any_ok = __any(any_ok);
                for (int idx = lane; idx < NN; idx += kWave) {
                    const int i = idx / N, j = idx % N;
                    const double gap = fabs(s_g[i] - s_g[j]);
                    const bool abs_ok = !any_ok || s_g[i] >= P.abs_gain_min_db || s_g[j] >= P.abs_gain_min_db;
                    const float hterm = P.score_w_history * s_hist[idx];       // float32 product
                    double S = P.score_w_delta_db * gap + (double)hterm;
                    if (!(s_feas[idx] && abs_ok)) S = -kInf;
                    if (P.qos_enable && !s_qos[idx] && finite(S)) S = S - P.qos_soft_penalty;
                    if (i == j) S = -kInf;
                    s_S[idx] = S;
                    s_ok[idx] = finite(S);             // = (feasible > 0) & isfinite(S)
                }
                __syncthreads();
                Ranks R;
                rank_entries(s_S, s_ok, NN, lane, R);
                busy = 0; mate = 0; npairs = 0; K_last = 0;
                if (R.cnt > 0) {
                    // ---- primary matching (TRAIN:326-398) -------------------------------------------
                    const double q = fmin(fmax(accept_q, 0.0), 1.0);
                    const double thr = quantile_linear(R, 1.0 - q);
                    for (int idx = lane; idx < NN; idx += kWave)
                        s_W[idx] = (s_ok[idx] && s_S[idx] >= thr) ? s_S[idx] : -kInf;
                    __syncthreads();
                    bool has_edge = false;
                    if (lane < N)
                        for (int u = 0; u < N; ++u)
                            if (u != lane) has_edge = has_edge || finite(s_W[min(lane, u) * N + max(lane, u)]);
                    unsigned live = (unsigned)__ballot(singles ? has_edge : lane < N) & 0xFFFFu;
                    const int K = __popc(live);
                    K_last = K;
                    if (K > 0) {
                        // compressed weights w[a][b], a < b (users in increasing order)
                        for (int idx = lane; idx < K * K; idx += kWave) {
                            const int a = idx / K, b = idx % K;
                            unsigned m = live;
                            int va = 0, vb = 0;
                            for (int c = 0, pos = 0; m; m &= m - 1, ++c) {
                                pos = __ffs(m) - 1;
                                if (c == a) va = pos;
                                if (c == b) vb = pos;
                            }
                            s_w[a * kNV + b] = a < b ? s_W[va * N + vb] : -kInf;
                        }
                        const int full = (1 << K) - 1;
                        double* T = s_dp;
                        if (K > KL) T = A.ns.scratch + ((size_t)blockIdx.x << N);
                        if (lane == 0) T[full] = 0.0;
                        for (int pc = K - 1; pc >= 0; --pc) {
                            __syncthreads();
                            for (int m = lane; m < full; m += kWave) {
                                if (__popc(m) != pc) continue;
                                int arg;
                                const double b = best_at(T, s_w, K, singles, m, arg);
                                T[m] = arg == -2 ? 0.0 : b;        // TRAIN:389-390
                            }
                        }
                        __syncthreads();
                        // walk the choices from the empty mask (every lane, same reads)
                        int m = 0;
                        while (m != full) {
                            int arg;
                            best_at(T, s_w, K, singles, m, arg);
                            if (arg == -2) break;
                            const int x = __ffs(~m) - 1;
                            m |= 1 << x;
                            if (arg >= 0) {
                                m |= 1 << arg;
                                unsigned lm = live;
                                int vx = 0, vj = 0;
                                for (int c = 0; lm; lm &= lm - 1, ++c) {
                                    const int pos = __ffs(lm) - 1;
                                    if (c == x) vx = pos;
                                    if (c == arg) vj = pos;
                                }
                                busy |= (1u << vx) | (1u << vj);
                                mate |= ((unsigned long long)vj << (4 * vx)) | ((unsigned long long)vx << (4 * vj));
                                ++npairs;
                            }
                        }
                    }
                    // ---- greedy completion (TRAIN:276-324) ------------------------------------------
                    if (npairs < target) {
                        const double thr2 = quantile_linear(R, P.completion_min_quantile);
                        while (npairs < target) {
                            double bs = -kInf;
                            int bi = -1;
                            for (int idx = lane; idx < NN; idx += kWave) {
                                const int i = idx / N, j = idx % N;
                                if (i < j && s_ok[idx] && s_S[idx] >= thr2 && !((busy >> i) & 1) && !((busy >> j) & 1)) {
                                    const double s = s_S[idx];
                                    if (bi < 0 || s > bs || (s == bs && idx > bi)) { bs = s; bi = idx; }
                                }
                            }
#pragma unroll
                            for (int o = kWave / 2; o > 0; o >>= 1) {
                                const double os = __shfl_xor(bs, o, kWave);
                                const int oi = __shfl_xor(bi, o, kWave);
                                if (oi >= 0 && (bi < 0 || os > bs || (os == bs && oi > bi))) { bs = os; bi = oi; }
                            }
                            if (bi < 0) break;
                            const int i = bi / N, j = bi % N;
                            busy |= (1u << i) | (1u << j);
                            mate |= ((unsigned long long)j << (4 * i)) | ((unsigned long long)i << (4 * j));
                            ++npairs;
                        }
                    }
                }
                // ---- back-off (TRAIN:1493-1524) -----------------------------------------------------
                if (npairs >= target || rounds >= P.mwm_backoff_rounds) break;
                ++rounds;
                K_back = min(N - 1, K_back + P.relax_topk_step);
                tau_b = fmax(P.tau_back_floor_db, tau_b * P.relax_tau_factor);
                __syncthreads();
                for (int idx = lane; idx < NN; idx += kWave) {     // _relax_mask_once, TRAIN:260-275
                    const int i = idx / N, j = idx % N;
                    const double gap = fabs(s_g[i] - s_g[j]);
                    int rank = 0;                                   // position in argsort(-gap[i]), equal keys by index
                    for (int k = 0; k < N; ++k) {
                        const double gk = fabs(s_g[i] - s_g[k]);
                        rank += (gk > gap || (gk == gap && k < j)) ? 1 : 0;
                    }
                    const bool top = K_back >= 1 && rank < min(K_back, N - 1);
                    const bool cand = gap >= tau_b && i != j;
                    s_feas[idx] = (s_feas[idx] || cand || top) ? 1 : 0;
                }
                accept_q = fmax(0.05, accept_q - P.mwm_accept_q_step);
                __syncthreads();
            }
            if (lane < N) {                            // episode_groups <- pairs + singles (TRAIN:1548-1553)
                int p = -1;
                if ((busy >> lane) & 1) {
                    const int m = (int)((mate >> (4 * lane)) & 15);
                    p = m > lane ? m : m + 65536;      // pairs are listed (low, high)
                }
                s_part[lane] = p;
                A.ns.partner[(long long)env * N + lane] = p;
            }
            flags |= RISVEC_NOMA_HAS_GROUPS;
            __syncthreads();
        }
        // ---- history / streak update (TRAIN:1556-1561), outputs --------------------------------------
        for (int idx = lane; idx < NN; idx += kWave) {
            const int i = idx / N, j = idx % N;
            const int p = s_part[i];
            float h = s_hist[idx];
            if (p >= 0 && (p & 0xFFFF) == j) h += 1.0f;
            hist[idx] = h;
        }
        if (lane < N) {
            const int p = s_part[lane];
            int* st = A.ns.streak + (long long)env * N + lane;
            *st = p >= 0 ? 0 : *st + 1;
            A.partner_out[(long long)env * N + lane] = p;
        }
        if (lane == 0) {
            const int ng = N - npairs;
            if (recompute) A.ns.n_groups[env] = ng;
            A.n_groups_out[env] = ng;
            A.ns.flags[env] = (uint8_t)flags;
            A.ns.last_global[env] = last;
            A.ns.best_global[env] = best;
            if (A.info_out) {
                int* o = A.info_out + (long long)env * 4;
                o[0] = recompute ? 1 : 0; o[1] = rounds; o[2] = npairs; o[3] = K_last;
            }
        }
    }
}

// tau = quantile q of |g_strong - g_weak| (TRAIN:842-855) and the feasibility mask (TRAIN:134-156).
__global__ void __launch_bounds__(kWave)
k_noma_mask(RisVecNomaState ns, const float* gain, const double* gdb15, double q_now, int K_now) {
    __shared__ double s_g[kNV], s_d[kNN];
    __shared__ uint8_t s_ok[kNN], s_m[kNN], s_keep[kNN];
    const int lane = threadIdx.x;
    const int N = ns.n_veh, NN = N * N;
    for (int env = blockIdx.x; env < ns.n_envs; env += gridDim.x) {
        __syncthreads();
        if (lane < N)
            s_g[lane] = gdb15 ? gdb15[(long long)env * N + lane]
                              : 10.0 * log10(fmax((double)gain[(long long)env * N + lane], 1e-15));
        __syncthreads();
        // weak half = the n/2 smallest (argsort, equal keys by index); diffs over strong x weak
        for (int idx = lane; idx < NN; idx += kWave) {
            const int i = idx / N, j = idx % N;
            int ri = 0, rj = 0;
            for (int k = 0; k < N; ++k) {
                ri += (s_g[k] < s_g[i] || (s_g[k] == s_g[i] && k < i)) ? 1 : 0;
                rj += (s_g[k] < s_g[j] || (s_g[k] == s_g[j] && k < j)) ? 1 : 0;
            }
            s_ok[idx] = ri >= N / 2 && rj < N / 2;          // i strong, j weak
            s_d[idx] = fabs(s_g[i] - s_g[j]);
        }
        __syncthreads();
        double tau = 0.0;
        if (N >= 2) {
            Ranks R;
            rank_entries(s_d, s_ok, NN, lane, R);
            tau = quantile_linear(R, q_now);
        }
        if (lane == 0) ns.tau[env] = tau;
        if (K_now < 1) continue;
        for (int idx = lane; idx < NN; idx += kWave) {
            const int i = idx / N, j = idx % N;
            s_m[idx] = (i != j && !(s_d[idx] < tau)) ? 1 : 0;
        }
        __syncthreads();
        for (int idx = lane; idx < NN; idx += kWave) {     // per-row top-K of the survivors
            const int i = idx / N, j = idx % N;
            int n_cand = 0, rank = 0;
            for (int k = 0; k < N; ++k) {
                if (!s_m[i * N + k]) continue;
                ++n_cand;
                const double gk = s_d[i * N + k];
                rank += (gk > s_d[idx] || (gk == s_d[idx] && k < j)) ? 1 : 0;
            }
            s_keep[idx] = s_m[idx] && (n_cand <= K_now || rank < K_now);
        }
        __syncthreads();
        for (int idx = lane; idx < NN; idx += kWave) {
            const int i = idx / N, j = idx % N;
            ns.mask[(long long)env * NN + idx] = s_keep[idx] && s_keep[j * N + i];
        }
    }
}

int noma_grid(int E, int cap) {
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 256;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
        return n;
    }();
    const int want = cus * 16;
    int g = E < want ? E : want;
    if (cap > 0 && g > cap) g = cap;
    return g < 1 ? 1 : g;
}

}  // namespace

hipError_t launch_noma_begin_episode(const RisVecNomaState& ns, hipStream_t st) {
    const size_t E = (size_t)ns.n_envs, N = (size_t)ns.n_veh;
    hipError_t err = hipMemsetAsync(ns.hist, 0, E * N * N * sizeof(float), st);
    if (err != hipSuccess) return err;
    err = hipMemsetAsync(ns.streak, 0, E * N * sizeof(int32_t), st);
    if (err != hipSuccess) return err;
    return hipMemsetAsync(ns.flags, 0, E, st);
}

hipError_t launch_noma_mask(const RisVecNomaState& ns, const float* gain, const double* gdb15, double q_now,
                            int K_now, hipStream_t st) {
    hipLaunchKernelGGL(k_noma_mask, dim3(noma_grid(ns.n_envs, 0)), dim3(kWave), 0, st, ns, gain, gdb15, q_now, K_now);
    return hipGetLastError();
}

// slots: how many envs may use the HBM spill of the matching table at once (0 = none available)
hipError_t launch_noma_group(const RisVecNomaState& ns, const RisVecNomaParams& p, const float* gain,
                             const double* gdb12, const float* p01, int use_mask, double q_back, int K_back,
                             const double* tau_back, const float* prev_global, int prev_stride, int i_step,
                             const float* u_unstick, uint64_t seed, uint32_t counter, int32_t* partner_out,
                             int32_t* n_groups_out, int32_t* info_out, int slots, hipStream_t st) {
    NomaArgs a{ns, p, gain, gdb12, p01, use_mask, q_back, K_back, tau_back, prev_global, prev_stride, i_step,
               u_unstick, seed, counter, partner_out, n_groups_out, info_out};
    if (ns.n_veh <= 8) {
        hipLaunchKernelGGL((k_noma_group<8>), dim3(noma_grid(ns.n_envs, 0)), dim3(kWave), 0, st, a);
    } else {
        const int cap = ns.n_veh > 12 ? slots : 0;     // every resident env needs its own spill slot
        hipLaunchKernelGGL((k_noma_group<12>), dim3(noma_grid(ns.n_envs, cap)), dim3(kWave), 0, st, a);
    }
    return hipGetLastError();
}

}  // namespace risvec
