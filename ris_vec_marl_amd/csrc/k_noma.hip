// f2: the NOMA grouping stage of the reference driver, marl_train_bcd.py (TRAIN) --
// the work done for every env right before env.step(): |delta g_dB| feasibility mask
// (TRAIN:128-156, 842-855), score matrix (TRAIN:164-194), quantile-gated max-weight matching
// (TRAIN:326-398), greedy completion (TRAIN:276-324), mask relaxation (TRAIN:260-275), pair QoS
// check (TRAIN:858-880) and the control logic with the freeze-in-episode safeties
// (TRAIN:1401-1562, 1618-1623).
//
// Almost every step of an episode is a FROZEN step, so k_noma_group (one launch per group() call) is
// built around that: a wavefront owns 8 consecutive envs; 8 lanes first do the per-env bookkeeping
// (reward tracking TRAIN:1618-1623, the freeze decision TRAIN:1527-1540); frozen envs only bump
// `pending` -- the history decay / pair increments and streak updates they owe are replayed,
// operation for operation, the next time somebody needs them (k_noma_flush for readers of those
// tensors) -- and the wavefront then solves, ONE ENV AT A TIME WITH ALL 64 LANES, the envs that
// asked for it.  A frozen step is a single short launch; when every env re-solves there are E/8
// wavefronts with 8 solves each, which still fills the chip.
// In a solve the N x N matrices (N <= 16) live in LDS with the entries dealt round-robin to
// the lanes; every decision the reference takes by comparing float64 numbers is taken by comparing
// float64 numbers formed in the same association order (fp contraction is off for the whole file:
// the matcher's exact ties are decided by last-bit rounding, see EXPERIMENTS.md, section 8 f2).
//   * quantiles: no sort -- each lane ranks its own entries against all (LDS broadcast reads) and
//     the two order statistics NumPy's linear method interpolates are picked by rank;
//   * matching: the reference's memoised recursion on the lowest unused user x, evaluated bottom-up
//     in layers of x (a state only needs states with a larger x), after dropping users without any
//     admissible edge (they pass the value through unchanged).  Only states the recursion can reach
//     are stored: with x the lowest unused user at most x users above it are taken, which leaves
//     2 583 of the 65 536 masks at 16 users; they are indexed by (x, size, colex rank) so the table
//     fits LDS for every K <= 16;
//   * completion: repeated wave-wide arg-max over the still-free admissible edges, which is what
//     the reference's sorted greedy scan selects.
// This is integer / branchy float64 work of a few hundred bytes per env; it is latency-bound.
#include "risvec_launch.hpp"

#pragma clang fp contract(off)

namespace risvec {
namespace {

constexpr int kNV = RISVEC_NOMA_MAX_VEH;
constexpr uint32_t kSiteUnstick = 7;                // Philox site of the TRAIN:1539 draw
constexpr double kInf = __builtin_huge_val();
constexpr int kBinW = 9;                            // binomials C(c, i), c < 16, i <= 8

struct NomaArgs {
    RisVecNomaState ns;
    RisVecNomaParams P;
    const float* gain;
    const double* gdb12;
    const float* p01;
    int p01_raw;                 // p01 is the raw SAC power head [E,N,2] in [-1,1]: apply TRAIN:1391-1396 when read
    int use_mask;
    int K_back;
    const double* tau_back;
    const float* prev_global;
    int prev_stride;
    int i_step;
    const float* u_unstick;
    uint64_t seed;
    uint32_t counter;
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
__device__ __forceinline__ bool paired_with(int p, int j) { return p >= 0 && (p & 0xFFFF) == j; }

// One deferred frozen step on a history entry / a streak (TRAIN:1406, 1556-1561).
__device__ __forceinline__ float replay_hist(float h, int pending, float decay, bool pair) {
    for (int k = 0; k < pending; ++k) {
        h = h * decay;
        if (pair) h += 1.0f;
    }
    return h;
}

// ---------------------------------------------------------------------------------------------
// bookkeeping of one env (run by one lane)
// ---------------------------------------------------------------------------------------------
// Per-env bookkeeping + freeze decision; returns true when the env has to (re)solve its pairing
// (`flags_out` = its updated flag bits, handed to the solving wavefront through a shuffle).
__device__ __forceinline__ bool noma_pre_env(const NomaArgs& A, int env, int& flags_out) {
    const RisVecNomaParams& P = A.P;
    // every load of the frozen path up front: one memory round trip instead of five dependent ones (a frozen
    // step is nothing but this function, so its latency IS the step's cost)
    // (the two optional inputs select their ADDRESS, not the load: a load under `if (pointer)` is merged with the
    // block that consumes it, and the loads behind it then wait for that block -- three round trips instead of one)
    const float* prev_p = A.prev_global ? A.prev_global + (long long)env * A.prev_stride
                                        : reinterpret_cast<const float*>(A.ns.pending + env);
    const float* u_p = A.u_unstick ? A.u_unstick + env : reinterpret_cast<const float*>(A.ns.pending + env);
    int flags = A.ns.flags[env];
    double last = A.ns.last_global[env], best = A.ns.best_global[env];
    const float prev = *prev_p;
    const int pend = A.ns.pending[env];
    const int n_groups = A.ns.n_groups[env];
    const float u_in = *u_p;
    if (A.prev_global) {                               // TRAIN:1618-1623, for the step that just ran
        const double g = (double)prev;
        if (!(flags & RISVEC_NOMA_HAS_LAST)) best = g;
        else if (g > best) best = g;
        last = g;
        flags |= RISVEC_NOMA_HAS_LAST;
        A.ns.last_global[env] = last;
        A.ns.best_global[env] = best;
    }
    const bool frozen = P.freeze_group_in_episode && (flags & RISVEC_NOMA_HAS_GROUPS);
    bool need_repair = false;
    if (frozen) {                                      // TRAIN:1527-1540
        if (P.freeze_recalc_every > 0 && A.i_step % P.freeze_recalc_every == 0) need_repair = true;
        if (!need_repair && (flags & RISVEC_NOMA_HAS_LAST) && !(flags & RISVEC_NOMA_UNSTICK_USED)) {
            if (last < best * (1.0 - P.freeze_reward_drop_ratio)) {
                need_repair = true;
                flags |= RISVEC_NOMA_UNSTICK_USED;
            }
        }
        if (!need_repair && P.freeze_unstick_prob > 0.0) {
            const double u = A.u_unstick
                ? (double)u_in
                : (double)u01(philox4x32_10((uint32_t)(A.ns.env_offset + env), 0u, A.counter, kSiteUnstick, A.seed).x);
            if (u < P.freeze_unstick_prob) need_repair = true;
        }
    }
    A.ns.flags[env] = (uint8_t)flags;
    flags_out = flags;
    if (frozen && !need_repair) {                      // reuse episode_groups (TRAIN:1542-1547): defer the bookkeeping
        A.ns.pending[env] = pend + 1;
        if (A.info_out) {
            int* o = A.info_out + (long long)env * 4;
            o[0] = 0; o[1] = 0; o[2] = A.ns.n_veh - n_groups; o[3] = 0;
        }
        return false;
    }
    return true;
}

// ---------------------------------------------------------------------------------------------
// flush: materialise the deferred frozen steps (one lane per matrix entry)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_noma_flush(RisVecNomaState ns, float decay) {
    const int N = ns.n_veh, NN = N * N;
    const long long gid = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= (long long)ns.n_envs * NN) return;
    const int env = (int)(gid / NN), idx = (int)(gid % NN), i = idx / N, j = idx % N;
    const int pend = ns.pending[env];
    if (pend == 0) return;
    const int p = ns.partner[(long long)env * N + i];
    ns.hist[gid] = replay_hist(ns.hist[gid], pend, decay, paired_with(p, j));
    if (j == 0) {
        int* st = ns.streak + (long long)env * N + i;
        *st = p >= 0 ? 0 : *st + pend;
    }
}
__global__ void __launch_bounds__(kBlock)
k_noma_clear_pending(RisVecNomaState ns) {
    const int env = blockIdx.x * kBlock + threadIdx.x;
    if (env < ns.n_envs) ns.pending[env] = 0;
}

// ---------------------------------------------------------------------------------------------
// group: a wavefront per 8 envs -- bookkeeping by 8 lanes, then one full-wave solve per env that needs it
// ---------------------------------------------------------------------------------------------
template <int NMAX>
struct Shape {
    static constexpr int NN = NMAX * NMAX;
    static constexpr int EPL = NN / kWave;             // matrix entries per lane
    static constexpr int DP = NMAX <= 8 ? 256 : 2600;  // matching table: 2^8 masks / the 2 583 reachable states at 16
    static constexpr int KPLAIN = NMAX <= 8 ? 8 : 11;  // up to here the table is simply indexed by the mask
};

template <int EPL>
struct Ranks {
    double x[EPL];
    int less[EPL], leq[EPL];
    int cnt;
};

// val[k] = +inf for entries outside the multiset, so they rank after everything.
template <int EPL>
__device__ __forceinline__ void rank_entries(const double* val, int n, int lane, Ranks<EPL>& R) {
    int mine = 0;
#pragma unroll
    for (int t = 0; t < EPL; ++t) {
        const int idx = lane + t * kWave;
        R.x[t] = idx < n ? val[idx] : kInf;
        R.less[t] = R.leq[t] = 0;
        mine += R.x[t] < kInf ? 1 : 0;
    }
#pragma unroll 8
    for (int k = 0; k < n; ++k) {                     // broadcast reads: every lane the same address
        const double y = val[k];
#pragma unroll
        for (int t = 0; t < EPL; ++t) {
            R.less[t] += y < R.x[t] ? 1 : 0;
            R.leq[t] += y <= R.x[t] ? 1 : 0;
        }
    }
    R.cnt = wave_sum(mine);
}

template <int EPL>
__device__ __forceinline__ double order_stat(const Ranks<EPL>& R, int k) {   // k-th smallest, 0-based
    double v = -kInf;
#pragma unroll
    for (int t = 0; t < EPL; ++t)
        if (R.x[t] < kInf && R.less[t] <= k && k < R.leq[t]) v = R.x[t];
    return wave_max(v);
}

// np.quantile(values, q), method 'linear': virtual index (n-1) q, neighbours floor / floor+1 (both
// the last element from n-1 up), two-sided lerp (a + d t below t = 0.5, b - d (1-t) from there).
template <int EPL>
__device__ __forceinline__ double quantile_linear(const Ranks<EPL>& R, double q) {
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

// Matching table: value of the recurrence at a reachable mask.  x = lowest unused user, T = users
// taken above x (|T| <= x); slot = base[x] + (number of smaller subsets of the K-1-x upper users)
// + colex rank of T among the subsets of its size.
struct MatchTab {
    const double* dp;
    const int* base;       // [K+1]
    const int* sizeoff;    // [16][kBinW + 1]
    const uint16_t* clo;   // [256]      colex rank of the low 8 bits of T
    const uint16_t* chi;   // [9][128]   colex rank contribution of bits 8.. of T, given popcount(low 8)
    int K, full;
    bool plain;            // 2^K fits the table: slot = mask (no ranking arithmetic on the critical path)
    __device__ __forceinline__ int slot(int mask) const {
        if (plain) return mask;
        const int x = __ffs(~mask) - 1;
        unsigned T = (unsigned)mask >> (x + 1);
        const int n = K - 1 - x;
        const unsigned lo = T & 255u, hi = T >> 8;
        return base[x] + sizeoff[n * (kBinW + 1) + __popc(T)] + clo[lo] + chi[__popc(lo) * 128 + hi];
    }
    __device__ __forceinline__ double value(int mask) const { return mask == full ? 0.0 : dp[slot(mask)]; }
};

// Value and choice of the recurrence at `mask` (TRAIN:360-391): the lowest unused user x stays single
// (arg -1) or takes partner j (arg j); arg -2 = no option (value 0, nothing below it).
// `adj[x]`: bit j set when (x, j), j > x, is an admissible edge -- the partners are walked in increasing j
// by peeling set bits, so a sparse graph (the usual case: only the top-quantile edges survive) costs a
// couple of iterations per state instead of one per user.
__device__ __forceinline__ double best_at(const MatchTab& M, const double* w, const int* adj, bool singles, int mask,
                                          int& arg) {
    const int x = __ffs(~mask) - 1;
    double best = -kInf;
    arg = -2;
    if (singles) {
        const double w1 = M.value(mask | (1 << x));
        if (w1 > best) { best = w1; arg = -1; }
    }
    for (unsigned cand = (unsigned)adj[x] & ~(unsigned)mask; cand; cand &= cand - 1) {
        const int j = __ffs(cand) - 1;
        const double we = w[x * kNV + j];
        const double w2 = M.value(mask | (1 << x) | (1 << j));
        if (finite(w2) && we + w2 > best) { best = we + w2; arg = j; }
    }
    return best;
}

// Envs per wavefront: as many wavefronts as the chip keeps resident when every env solves (8 users: ~20
// wavefronts per CU fit -> 8 envs each at 32 768 envs; 16 users: the 32 KB of LDS allow ~4 per CU -> 32
// envs each), so a frozen step launches no more blocks than that.
template <int NMAX> struct EnvsPerWave { static constexpr int value = NMAX <= 8 ? 8 : 32; };

// (One long function on purpose: split into inlined helpers over a shared-memory struct the same code ran the
// 16-user kernel's frozen path 4x slower and its solves 5 % slower, A/B on one box -- the compiler's schedule
// of this kernel is that sensitive to its shape.)
template <int NMAX>
__global__ void __launch_bounds__(kWave, NMAX <= 8 ? 5 : 1)
k_noma_group(NomaArgs A) {
    constexpr int kEnvsPerWave = EnvsPerWave<NMAX>::value;
    using S = Shape<NMAX>;
    constexpr int EPL = S::EPL;
    __shared__ double s_S[S::NN], s_R[S::NN], s_W[S::NN], s_w[kNV * kNV], s_g[kNV], s_lin[kNV], s_p[kNV];
    __shared__ double s_dp[S::DP];
    __shared__ float s_hist[S::NN];
    __shared__ uint8_t s_feas[S::NN], s_qos[S::NN];
    __shared__ int s_part[kNV], s_base[kNV + 1], s_adj[kNV], s_binom[16 * kBinW], s_sizeoff[16 * (kBinW + 1)];
    __shared__ signed char s_arg[S::DP];               // choice taken at each state, for the walk-back
    constexpr bool kRanked = NMAX > S::KPLAIN;         // more users than the plain 2^K table can hold?
    __shared__ uint16_t s_clo[kRanked ? 256 : 1], s_chi[kRanked ? 9 * 128 : 1];   // colex-rank lookup
    // a frozen step is a handful of loads and stores per env: every argument it touches in ONE scalar round trip
    RISVEC_ARGS_IN_ONE_TRIP("s"(A.ns.n_envs), "s"(A.ns.n_veh), "s"(A.ns.flags), "s"(A.ns.last_global), "s"(A.ns.best_global),
                            "s"(A.prev_global), "s"(A.prev_stride), "s"(A.ns.pending), "s"(A.ns.n_groups), "s"(A.u_unstick),
                            "s"(A.P.freeze_group_in_episode), "s"(A.P.freeze_recalc_every), "s"(A.i_step),
                            "s"(A.P.freeze_reward_drop_ratio), "s"(A.P.freeze_unstick_prob), "s"(A.info_out));
    const int lane = threadIdx.x;
    const int N = A.ns.n_veh, NN = N * N;
    const RisVecNomaParams& P = A.P;
    const bool singles = P.mwm_allow_singles != 0;

    int ei[EPL], ej[EPL];                              // this lane's matrix entries
    bool ein[EPL];
    bool tables_ready = false;
    // The common case -- a wavefront whose (only) env group is frozen -- is decided and LEAVES here, in front of the loop.
    // Whatever the solve path keeps loop-invariant (the Philox key schedule of the unstick draw, comparisons of its
    // parameters, the spills they cause: ~280 instructions and eight scalar waits in the ISA) is hoisted into the
    // loop's preheader, and with the check inside the loop that preheader ran before every frozen step's first load.
    const int e_first = blockIdx.x * kEnvsPerWave;
    if (e_first >= A.ns.n_envs) return;
    int flags_first = 0;
    bool solve_first = false;
    if (lane < kEnvsPerWave && e_first + lane < A.ns.n_envs) solve_first = noma_pre_env(A, e_first + lane, flags_first);
    const unsigned todo_first = (unsigned)__ballot(solve_first);
    if (todo_first == 0 && (long long)e_first + (long long)gridDim.x * kEnvsPerWave >= A.ns.n_envs) return;
    for (int e0 = e_first; e0 < A.ns.n_envs; e0 += gridDim.x * kEnvsPerWave) {
    int my_flags = flags_first;
    bool my_solve = solve_first;
    unsigned todo = todo_first;
    if (e0 != e_first) {
        my_flags = 0;
        my_solve = false;
        if (lane < kEnvsPerWave && e0 + lane < A.ns.n_envs) my_solve = noma_pre_env(A, e0 + lane, my_flags);
        todo = (unsigned)__ballot(my_solve);
    }
    if (todo == 0) continue;                           // all of the group's envs frozen
    if (!tables_ready) {
    tables_ready = true;
    // Only the (rare) wavefronts that solve anything need the tables below; the opaque move keeps the compiler
    // from hoisting their arithmetic into a prologue every frozen-step wavefront would then pay for.
    int tl = lane;
    asm volatile("" : "+v"(tl));
    // binomials C(c, i) and their prefix sums over i (Pascal rows; one row per lane)
    if (tl < 16) {
        int c = 1;                                     // C(lane, 0)
        int off = 0;
        for (int i = 0; i <= kBinW; ++i) {
            if (i < kBinW) s_binom[tl * kBinW + i] = c;
            s_sizeoff[tl * (kBinW + 1) + i] = off;
            off += c;
            c = i < tl ? c * (tl - i) / (i + 1) : 0;       // C(lane, i+1)
        }
    }
#pragma unroll
    for (int t = 0; t < EPL; ++t) {
        const int idx = tl + t * kWave;
        ein[t] = idx < NN;
        ei[t] = ein[t] ? idx / N : 0;
        ej[t] = ein[t] ? idx % N : 0;
    }
    if constexpr (kRanked) {                           // ranked table index: only beyond 2^KPLAIN masks
        __syncthreads();
        for (int b = lane; b < 256; b += kWave) {      // sum over set bits c_1 < c_2 < ... of C(c_i, i)
            int r = 0, i = 1;
            for (unsigned m = b; m; m &= m - 1, ++i) r += i < kBinW ? s_binom[(__ffs(m) - 1) * kBinW + i] : 0;
            s_clo[b] = (uint16_t)r;
        }
        for (int e = lane; e < 9 * 128; e += kWave) {  // bits 8.. (positions 8 + c), ranks continue at p + 1
            const int p = e / 128;
            int r = 0, i = p + 1;
            for (unsigned m = e % 128; m; m &= m - 1, ++i) r += i < kBinW ? s_binom[(8 + __ffs(m) - 1) * kBinW + i] : 0;
            s_chi[e] = (uint16_t)r;
        }
    }
    }   // tables
    for (; todo; todo &= todo - 1) {
        const int slot = __ffs(todo) - 1;
        const int env = e0 + slot;
        __syncthreads();                               // LDS reuse across envs
        const int flags = __shfl(my_flags, slot, kWave);
        const int pend = A.ns.pending[env];
        const bool had_groups = (flags & RISVEC_NOMA_HAS_GROUPS) != 0;
        float* hist = A.ns.hist + (long long)env * NN;
        if (lane < N) {
            const double g = (double)A.gain[(long long)env * N + lane];
            s_lin[lane] = g;
            s_g[lane] = A.gdb12 ? A.gdb12[(long long)env * N + lane] : 10.0 * log10(fmax(g, 1e-12));
            float p = 0.0f;
            if (A.p01) {
                if (A.p01_raw) {                               // TRAIN:1391-1396, as risvec_marshal_actions computes it
                    const float a0 = A.p01[((long long)env * N + lane) * 2];
                    p = (fminf(fmaxf(a0, -0.999f), 0.999f) + 1.0f) / 2.0f;
                } else {
                    p = A.p01[(long long)env * N + lane];
                }
            }
            s_p[lane] = (double)p;
            s_part[lane] = had_groups ? A.ns.partner[(long long)env * N + lane] : -1;
        }
        __syncthreads();
        // deferred frozen steps with the OLD groups, then this step's decay (TRAIN:1406)
#pragma unroll
        for (int t = 0; t < EPL; ++t) {
            if (!ein[t]) continue;
            const int idx = lane + t * kWave;
            float h = replay_hist(hist[idx], pend, P.pair_hist_decay, had_groups && paired_with(s_part[ei[t]], ej[t]));
            s_hist[idx] = h * P.pair_hist_decay;
            s_feas[idx] = (P.mask_enable && A.use_mask) ? A.ns.mask[(long long)env * NN + idx] : (ei[t] != ej[t]);
        }
        int streak = 0;
        if (lane < N) {
            streak = A.ns.streak[(long long)env * N + lane];
            if (pend > 0) streak = s_part[lane] >= 0 ? 0 : streak + pend;
        }
        if (P.qos_enable) {                            // TRAIN:1426-1441 + 858-880
#pragma unroll
            for (int t = 0; t < EPL; ++t) {
                if (!ein[t]) continue;
                const int i = ei[t], j = ej[t];
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
                s_qos[lane + t * kWave] = okq;
            }
        }
        // ================= solve (TRAIN:1419-1524) ==================================================
        const int target = max(1, P.min_pair_target);
        double accept_q = P.mwm_accept_quantile;
        int K_back = A.K_back;
        double tau_b = A.tau_back[env];
        unsigned busy = 0;                             // wave-uniform: users already paired
        unsigned long long mate = 0;                   // 4 bits per user, valid where busy
        int rounds = 0, npairs = 0, K_last = 0;
        while (true) {
            // ---- score matrix (TRAIN:164-194) ---------------------------------------------------
            bool any_ok = false;
#pragma unroll
            for (int t = 0; t < EPL; ++t) {
                if (!ein[t]) continue;
                const bool abs_ok = s_g[ei[t]] >= P.abs_gain_min_db || s_g[ej[t]] >= P.abs_gain_min_db;
                any_ok = any_ok || (s_feas[lane + t * kWave] && abs_ok);
            }
            any_ok = __any(any_ok);
#pragma unroll
            for (int t = 0; t < EPL; ++t) {
                if (!ein[t]) continue;
                const int idx = lane + t * kWave, i = ei[t], j = ej[t];
                const double gap = fabs(s_g[i] - s_g[j]);
                const bool abs_ok = !any_ok || s_g[i] >= P.abs_gain_min_db || s_g[j] >= P.abs_gain_min_db;
                const float hterm = P.score_w_history * s_hist[idx];       // float32 product
                double Sv = P.score_w_delta_db * gap + (double)hterm;
                if (!(s_feas[idx] && abs_ok)) Sv = -kInf;
                if (P.qos_enable && !s_qos[idx] && finite(Sv)) Sv = Sv - P.qos_soft_penalty;
                if (i == j) Sv = -kInf;
                s_S[idx] = Sv;
                s_R[idx] = finite(Sv) ? Sv : kInf;     // finite <=> (feasible > 0) & isfinite(S)
            }
            __syncthreads();
            Ranks<EPL> R;
            rank_entries<EPL>(s_R, NN, lane, R);
            busy = 0; mate = 0; npairs = 0; K_last = 0;
            if (R.cnt > 0) {
                // ---- primary matching (TRAIN:326-398) -----------------------------------------------
                const double q = fmin(fmax(accept_q, 0.0), 1.0);
                const double thr = quantile_linear<EPL>(R, 1.0 - q);
#pragma unroll
                for (int t = 0; t < EPL; ++t) {
                    if (!ein[t]) continue;
                    const int idx = lane + t * kWave;
                    s_W[idx] = (finite(s_S[idx]) && s_S[idx] >= thr) ? s_S[idx] : -kInf;
                }
                __syncthreads();
                bool has_edge = false;
                if (lane < N)
                    for (int u = 0; u < N; ++u)
                        if (u != lane) has_edge = has_edge || finite(s_W[min(lane, u) * N + max(lane, u)]);
                const unsigned live = (unsigned)__ballot(singles ? has_edge : lane < N) & 0xFFFFu;
                const int K = __popc(live);
                K_last = K;
                if (K > 0) {
                    // compressed weights w[a][b], a < b (users in increasing order); table layout
                    for (int idx = lane; idx < K * K; idx += kWave) {
                        const int a = idx / K, b = idx % K;
                        unsigned m = live;
                        int va = 0, vb = 0;
                        for (int c = 0; m; m &= m - 1, ++c) {
                            const int pos = __ffs(m) - 1;
                            if (c == a) va = pos;
                            if (c == b) vb = pos;
                        }
                        s_w[a * kNV + b] = a < b ? s_W[va * N + vb] : -kInf;
                    }
                    __syncthreads();
                    if (lane < K) {                             // admissible partners above each user
                        int bits = 0;
                        for (int b = lane + 1; b < K; ++b) bits |= finite(s_w[lane * kNV + b]) ? (1 << b) : 0;
                        s_adj[lane] = bits;
                    }
                    if (lane == 0) {
                        int off = 0;
                        for (int x = 0; x < K; ++x) {
                            s_base[x] = off;
                            const int n = K - 1 - x;
                            off += s_sizeoff[n * (kBinW + 1) + min(x, n) + 1];
                        }
                        s_base[K] = off;
                    }
                    __syncthreads();
                    const MatchTab MT{s_dp, s_base, s_sizeoff, s_clo, s_chi, K, (1 << K) - 1, K <= S::KPLAIN};
                    for (int x = K - 1; x >= 0; --x) {          // a state only needs states with a larger x
                        const int n = K - 1 - x, low = (1 << x) - 1;
                        for (int T = lane; T < (1 << n); T += kWave) {
                            if (__popc(T) > x) continue;         // not reachable
                            const int m = low | (T << (x + 1));
                            int arg;
                            const double b = best_at(MT, s_w, s_adj, singles, m, arg);
                            const int sl = MT.slot(m);
                            s_dp[sl] = arg == -2 ? 0.0 : b;            // TRAIN:389-390
                            s_arg[sl] = (signed char)arg;
                        }
                        __syncthreads();
                    }
                    // walk the choices from the empty mask (every lane, same reads)
                    int m = 0;
                    while (m != MT.full) {
                        const int arg = s_arg[MT.slot(m)];
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
                // ---- greedy completion (TRAIN:276-324) ----------------------------------------------
                if (npairs < target) {
                    const double thr2 = quantile_linear<EPL>(R, P.completion_min_quantile);
                    while (npairs < target) {
                        double bs = -kInf;
                        int bi = -1;
#pragma unroll
                        for (int t = 0; t < EPL; ++t) {
                            if (!ein[t]) continue;
                            const int idx = lane + t * kWave, i = ei[t], j = ej[t];
                            const double s = s_S[idx];
                            if (i < j && finite(s) && s >= thr2 && !((busy >> i) & 1) && !((busy >> j) & 1))
                                if (bi < 0 || s > bs || (s == bs && idx > bi)) { bs = s; bi = idx; }
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
            // ---- back-off (TRAIN:1493-1524) ---------------------------------------------------------
            if (npairs >= target || rounds >= P.mwm_backoff_rounds) break;
            ++rounds;
            K_back = min(N - 1, K_back + P.relax_topk_step);
            tau_b = fmax(P.tau_back_floor_db, tau_b * P.relax_tau_factor);
            __syncthreads();
#pragma unroll
            for (int t = 0; t < EPL; ++t) {                     // _relax_mask_once, TRAIN:260-275
                if (!ein[t]) continue;
                const int idx = lane + t * kWave, i = ei[t], j = ej[t];
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
        __syncthreads();
        // ---- episode_groups <- pairs + singles (TRAIN:1548-1553); history / streak (TRAIN:1556-1561) -----
        if (lane < N) {
            int p = -1;
            if ((busy >> lane) & 1) {
                const int m = (int)((mate >> (4 * lane)) & 15);
                p = m > lane ? m : m + 65536;          // pairs are listed (low, high)
            }
            s_part[lane] = p;
            A.ns.partner[(long long)env * N + lane] = p;
            A.ns.streak[(long long)env * N + lane] = p >= 0 ? 0 : streak + 1;
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < EPL; ++t) {
            if (!ein[t]) continue;
            const int idx = lane + t * kWave;
            float h = s_hist[idx];
            if (paired_with(s_part[ei[t]], ej[t])) h += 1.0f;
            hist[idx] = h;
        }
        if (lane == 0) {
            A.ns.n_groups[env] = N - npairs;
            A.ns.flags[env] = (uint8_t)(flags | RISVEC_NOMA_HAS_GROUPS);
            A.ns.pending[env] = 0;
            if (A.info_out) {
                int* o = A.info_out + (long long)env * 4;
                o[0] = 1; o[1] = rounds; o[2] = npairs; o[3] = K_last;
            }
        }
    }
    }   // 8-env groups
}

// tau = quantile q of |g_strong - g_weak| (TRAIN:842-855) and the feasibility mask (TRAIN:134-156).
template <int NMAX>
__global__ void __launch_bounds__(kWave)
k_noma_mask(RisVecNomaState ns, const float* gain, const double* gdb15, double q_now, int K_now) {
    constexpr int NNM = NMAX * NMAX, EPL = NNM / kWave;
    __shared__ double s_g[kNV], s_d[NNM], s_R[NNM];
    __shared__ uint8_t s_m[NNM], s_keep[NNM];
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
            const double dgap = fabs(s_g[i] - s_g[j]);
            s_d[idx] = dgap;
            s_R[idx] = (ri >= N / 2 && rj < N / 2) ? dgap : kInf;     // i strong, j weak
        }
        __syncthreads();
        double tau = 0.0;
        if (N >= 2) {
            Ranks<EPL> R;
            rank_entries<EPL>(s_R, NN, lane, R);
            tau = quantile_linear<EPL>(R, q_now);
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

int noma_grid(int E) {
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 256;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
        return n;
    }();
    const int want = cus * 16;
    const int g = E < want ? E : want;
    return g < 1 ? 1 : g;
}

}  // namespace

hipError_t launch_noma_begin_episode(const RisVecNomaState& ns, hipStream_t st) {
    const size_t E = (size_t)ns.n_envs, N = (size_t)ns.n_veh;
    hipError_t err = hipMemsetAsync(ns.hist, 0, E * N * N * sizeof(float), st);
    if (err != hipSuccess) return err;
    err = hipMemsetAsync(ns.streak, 0, E * N * sizeof(int32_t), st);
    if (err != hipSuccess) return err;
    err = hipMemsetAsync(ns.pending, 0, E * sizeof(int32_t), st);
    if (err != hipSuccess) return err;
    return hipMemsetAsync(ns.flags, 0, E, st);
}

hipError_t launch_noma_mask(const RisVecNomaState& ns, const float* gain, const double* gdb15, double q_now,
                            int K_now, hipStream_t st) {
    const dim3 grid(noma_grid(ns.n_envs));
    if (ns.n_veh <= 8) hipLaunchKernelGGL(k_noma_mask<8>, grid, dim3(kWave), 0, st, ns, gain, gdb15, q_now, K_now);
    else hipLaunchKernelGGL(k_noma_mask<16>, grid, dim3(kWave), 0, st, ns, gain, gdb15, q_now, K_now);
    return hipGetLastError();
}

hipError_t launch_noma_flush(const RisVecNomaState& ns, float decay, hipStream_t st) {
    const long long n = (long long)ns.n_envs * ns.n_veh * ns.n_veh;
    hipLaunchKernelGGL(k_noma_flush, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, ns, decay);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(k_noma_clear_pending, dim3((ns.n_envs + kBlock - 1) / kBlock), dim3(kBlock), 0, st, ns);
    return hipGetLastError();
}

hipError_t launch_noma_group(const RisVecNomaState& ns, const RisVecNomaParams& p, const float* gain,
                             const double* gdb12, const float* p01, int p01_raw, int use_mask, int K_back,
                             const double* tau_back, const float* prev_global, int prev_stride, int i_step,
                             const float* u_unstick, uint64_t seed, uint32_t counter, int32_t* info_out,
                             hipStream_t st) {
    NomaArgs a{ns, p, gain, gdb12, p01, p01_raw, use_mask, K_back, tau_back, prev_global, prev_stride, i_step,
               u_unstick, seed, counter, info_out};
    const int epw = ns.n_veh <= 8 ? EnvsPerWave<8>::value : EnvsPerWave<16>::value;
    long long waves = ((long long)ns.n_envs + epw - 1) / epw;
    if (waves > (1 << 20)) waves = 1 << 20;            // grid-stride beyond that
    const dim3 grid((unsigned)waves);
    if (ns.n_veh <= 8) hipLaunchKernelGGL((k_noma_group<8>), grid, dim3(kWave), 0, st, a);
    else hipLaunchKernelGGL((k_noma_group<16>), grid, dim3(kWave), 0, st, a);
    return hipGetLastError();
}

}  // namespace risvec
