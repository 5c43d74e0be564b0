// f2: the NOMA grouping stage of the reference driver, marl_train_bcd.py (TRAIN) --
// the work done for every env right before env.step(): |delta g_dB| feasibility mask
// (TRAIN:128-156, 842-855), score matrix (TRAIN:164-194), quantile-gated max-weight matching
// (TRAIN:326-398), greedy completion (TRAIN:276-324), mask relaxation (TRAIN:260-275), pair QoS
// check (TRAIN:858-880) and the control logic with the freeze-in-episode safeties
// (TRAIN:1401-1562, 1618-1623).
//
// Almost every step of an episode is a FROZEN step, so k_noma_group (one launch per group() call up to 8
// users, two beyond: see Shape) is built around that: a wavefront owns 8 consecutive envs; 8 lanes first do the per-env bookkeeping
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
//   * quantiles: the wavefront sorts the scores with a bitonic network in registers (64 entries, one
//     per lane, up to 8 users; 256 entries, four per lane, beyond: ranking every entry against all took
//     n^2 float64 compares, a third of a 16-user solve) and the two order statistics NumPy's linear
//     method interpolates are read by rank (the mask kernel's quantile is over at most 64 gaps, 16 of
//     them at 8 users: there each lane still ranks its value against the others, which is cheaper);
//   * matching: the reference's memoised recursion on the lowest unused user x, evaluated bottom-up
//     in layers of x (a state only needs states with a larger x), after dropping users without any
//     admissible edge (they pass the value through unchanged).  Only states the recursion can reach
//     are stored.  Up to 8 users the table is simply indexed by the mask.  Beyond that it is indexed
//     by the FRONTIER: the users above x that can already be taken when x is the lowest unused one are
//     partners of somebody below x, F(x) = {j > x : (x', j) admissible for an x' < x}, so layer x holds
//     at most 2^|F(x)| states -- a few dozen in all for the sparse graphs the accept quantile leaves,
//     instead of 2^K; on the complete graph it is every mask the recursion could reach on any graph
//     (at most x users taken above x: 2 583 of the 65 536 masks at 16 users).  An env whose table
//     exceeds 256 states (about 1 % of them) is left to a second launch that has the LDS for it;
//   * completion: repeated wave-wide arg-max over the still-free admissible edges, which is what
//     the reference's sorted greedy scan selects.
// This is integer / branchy float64 work of a few hundred bytes per env; it is latency-bound, which
// is why the 16-user kernel's LDS is kept to 9 KB: four wavefronts per SIMD instead of one.
#include "risvec_launch.hpp"

#include <cstdlib>
#include <type_traits>

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
    void* scratch;               // Deferred (more than 8 users): the envs the first launch leaves to the second
    long long* stamps;           // diagnostic build only
    double qos_s_lo, qos_s_hi;   // (2^qos_R_min - 1) -/+ a 1e-9 band: outside it log2(1 + sinr) >= R_min is decided by a product
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
constexpr int kReachMax = 2583;                        // masks the recursion can reach at 16 users (at most x users taken above x)
constexpr int kTabMax = 2640;                          // >= the largest frontier table: kReachMax, + 49 when layer 1 stores 64
                                                       // states `direct` where the complete graph has 15 (see Layer)
// MODE 0: the kernel every group() call launches.  MODE 1 (more than 8 users only): the second launch, one wavefront per
// env whose matching table did not fit MODE 0's -- about 1 % of the envs (all gains at the floor: every score equal,
// every edge admitted) -- with LDS for the largest table there is.  Keeping that table out of MODE 0 is what lets it
// run four wavefronts per SIMD instead of one.
template <int NMAX, int MODE = 0>
struct Shape {
    static constexpr int NN = NMAX * NMAX;
    static constexpr int EPL = NN / kWave;             // matrix entries per lane
    static constexpr bool BIG = NMAX > 8;              // more than 8 users: frontier-indexed matching table, second launch
    static constexpr int DP = !BIG ? 256 : (MODE == 0 ? 256 : kTabMax);   // LDS table entries (up to 8 users: the 2^8 masks)
};
// Scratch of the two-launch scheme: the envs MODE 0 left for MODE 1.
struct Deferred {
    int count, done, pad[2];
    int env[1];                                        // [n_envs]
};
constexpr int kDeferredGridMax = 512;

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

// 256 values, four per lane, sorted ascending across the wavefront: position p is v[p & 3] of lane p >> 2.
// Bitonic network on element index e = 4 lane + t: the exchanges at distance 1 and 2 stay inside a lane, the
// others pair lane with lane ^ (distance / 4).  Ranking 256 entries against each other took 2 x 256 float64
// compares per entry (a third of a 16-user solve); this is 36 stages of one compare per entry.
template <int EPL>
struct Sorted {
    double v[EPL];
    int cnt;
};
using Sorted4 = Sorted<4>;
__device__ __forceinline__ void cmpx(double& a, double& b, bool asc) {
    const bool sw = asc ? b < a : a < b;
    const double lo = sw ? b : a, hi = sw ? a : b;
    a = lo;
    b = hi;
}
template <int D>
__device__ __forceinline__ void cmpx_lanes(double (&v)[4], int lane, bool asc) {
    const bool keep_min = ((lane & D) == 0) == asc;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const double other = __shfl_xor(v[t], D, kWave);
        const bool take = keep_min ? other < v[t] : v[t] < other;
        v[t] = take ? other : v[t];
    }
}
__device__ __forceinline__ void cmpx_in_lane(double (&v)[4], bool asc) {
    cmpx(v[0], v[2], asc);
    cmpx(v[1], v[3], asc);
    cmpx(v[0], v[1], asc);
    cmpx(v[2], v[3], asc);
}
__device__ __forceinline__ void sort256(double (&v)[4], int lane) {
    cmpx(v[0], v[1], true);                            // runs of 2: up, down
    cmpx(v[2], v[3], false);
    cmpx_in_lane(v, (lane & 1) == 0);                  // runs of 4
    bool asc = (lane & 2) == 0;                        // runs of 8
    cmpx_lanes<1>(v, lane, asc);
    cmpx_in_lane(v, asc);
    asc = (lane & 4) == 0;                             // 16
    cmpx_lanes<2>(v, lane, asc);
    cmpx_lanes<1>(v, lane, asc);
    cmpx_in_lane(v, asc);
    asc = (lane & 8) == 0;                             // 32
    cmpx_lanes<4>(v, lane, asc);
    cmpx_lanes<2>(v, lane, asc);
    cmpx_lanes<1>(v, lane, asc);
    cmpx_in_lane(v, asc);
    asc = (lane & 16) == 0;                            // 64
    cmpx_lanes<8>(v, lane, asc);
    cmpx_lanes<4>(v, lane, asc);
    cmpx_lanes<2>(v, lane, asc);
    cmpx_lanes<1>(v, lane, asc);
    cmpx_in_lane(v, asc);
    asc = (lane & 32) == 0;                            // 128
    cmpx_lanes<16>(v, lane, asc);
    cmpx_lanes<8>(v, lane, asc);
    cmpx_lanes<4>(v, lane, asc);
    cmpx_lanes<2>(v, lane, asc);
    cmpx_lanes<1>(v, lane, asc);
    cmpx_in_lane(v, asc);
    cmpx_lanes<32>(v, lane, true);                     // 256
    cmpx_lanes<16>(v, lane, true);
    cmpx_lanes<8>(v, lane, true);
    cmpx_lanes<4>(v, lane, true);
    cmpx_lanes<2>(v, lane, true);
    cmpx_lanes<1>(v, lane, true);
    cmpx_in_lane(v, true);
}
// 64 values, one per lane (up to 8 users), sorted ascending across the wavefront: 21 cross-lane stages.
template <int D>
__device__ __forceinline__ void cmpx_lane1(double& v, int lane, bool asc) {
    const bool keep_min = ((lane & D) == 0) == asc;
    const double other = __shfl_xor(v, D, kWave);
    const bool take = keep_min ? other < v : v < other;
    v = take ? other : v;
}
__device__ __forceinline__ void sort64(double& v, int lane) {
    cmpx_lane1<1>(v, lane, (lane & 2) == 0);
    bool asc = (lane & 4) == 0;
    cmpx_lane1<2>(v, lane, asc); cmpx_lane1<1>(v, lane, asc);
    asc = (lane & 8) == 0;
    cmpx_lane1<4>(v, lane, asc); cmpx_lane1<2>(v, lane, asc); cmpx_lane1<1>(v, lane, asc);
    asc = (lane & 16) == 0;
    cmpx_lane1<8>(v, lane, asc); cmpx_lane1<4>(v, lane, asc); cmpx_lane1<2>(v, lane, asc); cmpx_lane1<1>(v, lane, asc);
    asc = (lane & 32) == 0;
    cmpx_lane1<16>(v, lane, asc); cmpx_lane1<8>(v, lane, asc); cmpx_lane1<4>(v, lane, asc); cmpx_lane1<2>(v, lane, asc);
    cmpx_lane1<1>(v, lane, asc);
    cmpx_lane1<32>(v, lane, true); cmpx_lane1<16>(v, lane, true); cmpx_lane1<8>(v, lane, true); cmpx_lane1<4>(v, lane, true);
    cmpx_lane1<2>(v, lane, true); cmpx_lane1<1>(v, lane, true);
}
__device__ __forceinline__ double order_stat(const Sorted<1>& R, int k) { return __shfl(R.v[0], k, kWave); }
__device__ __forceinline__ void sort_wave(Sorted<1>& R, int lane) { sort64(R.v[0], lane); }
__device__ __forceinline__ void sort_wave(Sorted<4>& R, int lane) { sort256(R.v, lane); }
__device__ __forceinline__ double order_stat(const Sorted4& R, int k) {       // k wave-uniform
    // (the four values pass through opaque moves: a select between loads of R.v[] is otherwise folded into ONE load at a
    // selected address, and an array indexed at run time lives in scratch memory)
    double a = R.v[0], b = R.v[1], c = R.v[2], d = R.v[3];
    asm("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    double sel = a;
    if ((k & 3) == 1) sel = b;
    if ((k & 3) == 2) sel = c;
    if ((k & 3) == 3) sel = d;
    return __shfl(sel, k >> 2, kWave);
}

// np.quantile(values, q), method 'linear': virtual index (n-1) q, neighbours floor / floor+1 (both
// the last element from n-1 up), two-sided lerp (a + d t below t = 0.5, b - d (1-t) from there).
template <class RK>
__device__ __forceinline__ double quantile_linear(const RK& R, double q) {
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

// ---- matching tables: value / choice of the recurrence at a mask, two ways of finding its slot ---------------------
// (compressed users 0..K-1; x = lowest unused user of the mask, T = the users taken above x)
struct PlainTab {                                      // slot = mask
    double* dp;
    signed char* arg;
    int full;
    __device__ __forceinline__ int slot(int mask) const { return mask; }
    __device__ __forceinline__ double value(int mask) const { return mask == full ? 0.0 : dp[mask]; }
};

// Binomial sums for ranking subsets by (size, colex order): compile-time tables in constant memory, read only for the
// (rare) layers too wide to store every subset of their frontier.
struct ColexTab {
    int sizeoff[16 * (kBinW + 1)];                     // [c][i] = sum_{i' < i} C(c, i')
    uint16_t clo[256];                                 // colex rank of the low 8 bits of a subset
    uint16_t chi[9 * 128];                             // contribution of bits 8.., given popcount(low 8)
};
constexpr ColexTab make_colex() {
    ColexTab t{};
    int binom[16][kBinW] = {};                         // C(c, i), i < kBinW (larger i never ranks: at most 8 of <= 15 taken)
    for (int c = 0; c < 16; ++c)
        for (int i = 0; i < kBinW; ++i) binom[c][i] = i == 0 ? 1 : (c == 0 ? 0 : binom[c - 1][i - 1] + binom[c - 1][i]);
    for (int c = 0; c < 16; ++c) {
        int off = 0;
        for (int i = 0; i <= kBinW; ++i) {
            t.sizeoff[c * (kBinW + 1) + i] = off;
            if (i < kBinW) off += binom[c][i];
        }
    }
    for (int b = 0; b < 256; ++b) {                    // sum over set bits c_1 < c_2 < ... of C(c_i, i)
        int r = 0, i = 1;
        for (int c = 0; c < 8; ++c)
            if ((b >> c) & 1) { r += i < kBinW ? binom[c][i] : 0; ++i; }
        t.clo[b] = (uint16_t)r;
    }
    for (int e = 0; e < 9 * 128; ++e) {                // bits 8.. (positions 8 + c), ranks continue at p + 1
        int r = 0, i = e / 128 + 1;
        for (int c = 0; c < 7; ++c)
            if (((e % 128) >> c) & 1) { r += i < kBinW ? binom[8 + c][i] : 0; ++i; }
        t.chi[e] = (uint16_t)r;
    }
    return t;
}
__constant__ const ColexTab kColex = make_colex();
constexpr int reachable_masks(int K) {                 // table size of the complete graph on K users
    const ColexTab t = make_colex();
    int off = 0;
    for (int x = 0; x < K; ++x) {
        const int n = K - 1 - x;
        off += t.sizeoff[n * (kBinW + 1) + (x < n ? x : n) + 1];
    }
    return off;
}
static_assert(reachable_masks(16) == kReachMax && kTabMax >= kReachMax - 15 + 64, "kTabMax holds the largest table there is");

// Frontier-indexed table.  With x the lowest unused user, every user taken above x is the partner of somebody below x:
// T is a subset of F(x) = {j > x : (x', j) admissible for an x' < x}, of at most cap(x) = #{x' < x with an edge to above x}
// users.  Layer x stores those subsets: all 2^|F(x)| of them when that is at most a wavefront's worth (`direct`, slot =
// base[x] + T's bits gathered at the positions of F(x)), otherwise only the ones of size <= cap(x), ranked by (size, colex
// order).  On the sparse graphs the accept quantile leaves this is a few dozen states; on the complete graph it is the
// table of every mask the recursion can reach on any graph (2 583 states at 16 users).  No table is larger than that one
// by more than 49 states: a ranked layer holds at most what the complete graph's does (cap(x) <= x, |F(x)| <= 15 - x), and
// so does a direct one (<= 64 states) except layer 1, where the complete graph has 15.
struct Layer {
    unsigned long long pos;                            // 4 bits per user j: the rank of j in F(x)
    int base;                                          // first slot of the layer
    unsigned meta;                                     // F(x) | |F(x)| << 16 | cap(x) << 20 | direct << 24
};
__device__ __forceinline__ int rank_in_layer(const ColexTab* cx, unsigned meta, unsigned c) {   // c = T gathered at F(x)'s positions
    if (meta >> 24) return (int)c;
    const unsigned nf = (meta >> 16) & 15u, lo = c & 255u, hi = c >> 8;
    return cx->sizeoff[nf * (kBinW + 1) + __popc(c)] + cx->clo[lo] + cx->chi[__popc(lo) * 128 + hi];
}
struct FrontTab {
    double* dp;                                        // LDS: Shape::DP entries (256 in the first launch, kTabMax in the second)
    signed char* arg;
    const Layer* layer;                                // [K]
    const ColexTab* colex;                             // constant memory, or the second launch's copy in LDS
    int full, last;                                    // last = table size - 1
    __device__ __forceinline__ int slot(int mask) const {
        const int x = __ffs(~mask) - 1;
        const Layer L = layer[x];
        unsigned c = 0;
        for (unsigned m = (unsigned)mask & ~((2u << x) - 1u); m; m &= m - 1) c |= 1u << ((L.pos >> (4 * (__ffs(m) - 1))) & 15u);
        // (states that are stored but cannot be reached may ask for one that is not stored: whatever they read is never
        // used, but it has to be read inside the table)
        return min(L.base + rank_in_layer(colex, L.meta, c), last);
    }
    __device__ __forceinline__ double value(int mask) const { return mask == full ? 0.0 : dp[slot(mask)]; }
};

// solve_at for the frontier table, FOUR options at a time: their layer records are read together, then their table
// values, then they are folded in the reference's order (single first, partners by increasing j).  One option after the
// other is two dependent LDS round trips each; a state of the complete graph has a dozen.
__device__ __forceinline__ void solve_at_front(const FrontTab& M, const double* w, unsigned adj_x, bool singles, int x, int mask,
                                               int sl) {
    const int m1 = mask | (1 << x);
    double best = -kInf;
    int arg = -2;
    unsigned cand = adj_x & ~(unsigned)mask;
    bool single_pending = singles;
    do {
        int jj[4], xs[4];
        unsigned tt[4], cc[4];
        bool ok[4], is_full[4];
        Layer L[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool take = !(k == 0 && single_pending) && cand != 0;
            ok[k] = take || (k == 0 && single_pending);
            jj[k] = take ? __ffs(cand) - 1 : -1;
            if (take) cand &= cand - 1;
            int m = jj[k] >= 0 ? m1 | (1 << jj[k]) : m1;
            is_full[k] = m == M.full;
            if (is_full[k]) m = mask;                  // any stored state: its value is not used
            xs[k] = __ffs(~m) - 1;
            tt[k] = (unsigned)m & ~((2u << xs[k]) - 1u);
        }
        single_pending = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) L[k] = M.layer[xs[k]];
        unsigned rest = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned c = 0, m = tt[k];
#pragma unroll
            for (int r = 0; r < 3; ++r) {              // three taken users without a branch, the (rare) rest in a loop
                const int b = m ? __ffs(m) - 1 : 0;
                c |= m ? 1u << ((L[k].pos >> (4 * b)) & 15u) : 0u;
                m &= m - 1;
            }
            cc[k] = c;
            tt[k] = m;
            rest |= m;
        }
        if (rest) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                for (unsigned m = tt[k]; m; m &= m - 1) cc[k] |= 1u << ((L[k].pos >> (4 * (__ffs(m) - 1))) & 15u);
        }
        double v[4], we[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int s = min(L[k].base + rank_in_layer(M.colex, L[k].meta, cc[k]), M.last);
            v[k] = M.dp[s];
            we[k] = w[x * kNV + max(jj[k], 0)];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double vk = is_full[k] ? 0.0 : v[k];
            if (!ok[k]) continue;
            if (jj[k] < 0) {
                if (vk > best) { best = vk; arg = -1; }
            } else if (finite(vk) && we[k] + vk > best) {
                best = we[k] + vk;
                arg = jj[k];
            }
        }
    } while (cand);
    M.dp[sl] = arg == -2 ? 0.0 : best;                 // TRAIN:389-390
    M.arg[sl] = (signed char)arg;
}

// Value and choice of the recurrence at `mask` (TRAIN:360-391): the lowest unused user x stays single
// (arg -1) or takes partner j (arg j); arg -2 = no option (value 0, nothing below it).
// `adj[x]`: bit j set when (x, j), j > x, is an admissible edge -- the partners are walked in increasing j
// by peeling set bits, so a sparse graph (the usual case: only the top-quantile edges survive) costs a
// couple of iterations per state instead of one per user.
template <class Tab>
__device__ __forceinline__ void solve_at(const Tab& M, const double* w, unsigned adj_x, bool singles, int x, int mask, int sl) {
    double best = -kInf;
    int arg = -2;
    if (singles) {
        const double w1 = M.value(mask | (1 << x));
        if (w1 > best) { best = w1; arg = -1; }
    }
    for (unsigned cand = adj_x & ~(unsigned)mask; cand; cand &= cand - 1) {
        const int j = __ffs(cand) - 1;
        const double we = w[x * kNV + j];
        const double w2 = M.value(mask | (1 << x) | (1 << j));
        if (finite(w2) && we + w2 > best) { best = we + w2; arg = j; }
    }
    M.dp[sl] = arg == -2 ? 0.0 : best;                 // TRAIN:389-390
    M.arg[sl] = (signed char)arg;
}

// Layers of x from the top, table indexed by the mask: every mask with at most x users taken above x (up to 8 users).
__device__ __forceinline__ void solve_plain(const PlainTab& M, const double* w, const int* adj, bool singles, int K, int lane) {
    for (int x = K - 1; x >= 0; --x) {                 // a state only needs states with a larger x
        const int n = K - 1 - x, low = (1 << x) - 1;
        for (int T = lane; T < (1 << n); T += kWave) {
            if (__popc(T) > x) continue;               // not reachable
            const int m = low | (T << (x + 1));
            solve_at(M, w, (unsigned)adj[x], singles, x, m, m);
        }
        __syncthreads();
    }
}

// ... table indexed by the frontier.  `meta` / `base` / `adj`: lane x holds layer x's.  BATCH: four options at a time (the
// second launch: lone wavefronts on wide tables, all latency; in the first it only costs registers).
template <bool BATCH>
__device__ __forceinline__ void solve_frontier(const FrontTab& M, const double* w, unsigned meta, int base, unsigned adj,
                                               bool singles, int K, int lane) {
    for (int x = K - 1; x >= 0; --x) {
        const unsigned mx = (unsigned)__builtin_amdgcn_readlane((int)meta, x);
        const int bx = __builtin_amdgcn_readlane(base, x);
        const unsigned adj_x = (unsigned)__builtin_amdgcn_readlane((int)adj, x);
        const unsigned Fx = mx & 0xFFFFu;
        const int n_sub = 1 << ((mx >> 16) & 15u), cap = (int)((mx >> 20) & 15u), low = (1 << x) - 1;
        for (int c = lane; c < n_sub; c += kWave) {
            if (__popc(c) > cap) continue;             // more taken than users below x could have taken
            unsigned T = 0;
            int b = 0;
            for (unsigned m = Fx; m; m &= m - 1, ++b) T |= (((unsigned)c >> b) & 1u) << (__ffs(m) - 1);
            const int sl = bx + rank_in_layer(M.colex, mx, (unsigned)c);
            if constexpr (BATCH) solve_at_front(M, w, adj_x, singles, x, low | (int)T, sl);
            else solve_at(M, w, adj_x, singles, x, low | (int)T, sl);
        }
        __syncthreads();
    }
}

// Walk the choices from the empty mask (every lane, same reads) -> pairs in original user numbers.
template <class Tab>
__device__ __forceinline__ void walk_choices(const Tab& M, unsigned live, unsigned& busy, unsigned long long& mate,
                                             int& npairs) {
    unsigned long long user_of = 0;                    // 4 bits per compressed index: the user it stands for
    {
        int c = 0;
        for (unsigned lm = live; lm; lm &= lm - 1, ++c) user_of |= (unsigned long long)(__ffs(lm) - 1) << (4 * c);
    }
    int m = 0;
    while (m != M.full) {
        const int arg = M.arg[M.slot(m)];
        if (arg == -2) break;
        const int x = __ffs(~m) - 1;
        m |= 1 << x;
        if (arg >= 0) {
            m |= 1 << arg;
            const int vx = (int)((user_of >> (4 * x)) & 15u), vj = (int)((user_of >> (4 * arg)) & 15u);
            busy |= (1u << vx) | (1u << vj);
            mate |= ((unsigned long long)vj << (4 * vx)) | ((unsigned long long)vx << (4 * vj));
            ++npairs;
        }
    }
}

// Envs per wavefront: as many wavefronts as the chip keeps resident when every env solves (16 to 20 wavefronts
// per CU fit -> 8 envs each at 32 768 envs), so a frozen step launches no more blocks than that.
template <int NMAX> struct EnvsPerWave { static constexpr int value = 8; };

// (One long function on purpose: split into inlined helpers over a shared-memory struct the same code ran the
// 16-user kernel's frozen path 4x slower and its solves 5 % slower, A/B on one box -- the compiler's schedule
// of this kernel is that sensitive to its shape.)
// STAMP (diagnostic library only, tools/noma_stamps.py): s_memtime ticks per phase of the solves, summed per block into
// A.stamps.
constexpr int kStamps = 12;
#ifndef RISVEC_NOMA_OCC
#define RISVEC_NOMA_OCC 4                              // wavefronts per SIMD the 16-user first launch is compiled for
#endif
template <int NMAX, int MODE = 0, bool STAMP = false>
__global__ void __launch_bounds__(kWave, NMAX <= 8 ? 5 : (MODE == 0 ? RISVEC_NOMA_OCC : 1))
k_noma_group(NomaArgs A) {
    constexpr int kEnvsPerWave = EnvsPerWave<NMAX>::value;
    using S = Shape<NMAX, MODE>;
    constexpr int EPL = S::EPL;
    constexpr bool BIG = S::BIG;
    __shared__ double s_S[S::NN], s_w[kNV * kNV], s_g[kNV], s_lin[kNV], s_p[kNV];
    __shared__ double s_dp[S::DP];
    __shared__ float s_hist[S::NN];
    __shared__ uint8_t s_feas[S::NN], s_qos[S::NN];
    __shared__ int s_part[kNV], s_adj[kNV];
    __shared__ unsigned s_live[1];
    __shared__ Layer s_layer[BIG ? kNV : 1];
    using ColexLds = std::conditional_t<MODE == 1, ColexTab, int>;       // MODE 1 (its tables are the wide ones) keeps a copy in LDS
    __shared__ ColexLds s_colex;
    const ColexTab* colex = &kColex;
    if constexpr (MODE == 1) colex = &s_colex;
    __shared__ signed char s_arg[S::DP];               // choice taken at each state, for the walk-back
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
    long long t_acc[kStamps] = {}, t_last = 0;
#define RISVEC_TICK(i)                                                      \
    if constexpr (STAMP) {                                                  \
        const long long now = (long long)__builtin_amdgcn_s_memtime();      \
        t_acc[i] += now - t_last;                                           \
        t_last = now;                                                       \
    }
    Deferred* const deferred_list = static_cast<Deferred*>(A.scratch);
    // One env's pairing, all 64 lanes.  (A lambda with a single call site per instantiation: MODE 0 calls it for the envs its
    // bookkeeping lanes flagged, MODE 1 for the envs MODE 0 left in the list.)
    const auto solve_env = [&](const int env, const int flags) {
        if constexpr (STAMP) { t_last = (long long)__builtin_amdgcn_s_memtime(); t_acc[11] += 1; }
        __syncthreads();                               // LDS reuse across envs
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
            // log2(1 + num/den) >= R_min is num >= (2^R_min - 1) den except within rounding of the boundary: only a pair inside
            // a +-1e-9 band around it (never, in practice) pays for the float64 division and log2 -- a fifth of a solve with
            // the QoS check on
            const auto rate_ok = [&](double num, double den) {
                num = fmax(num, 0.0);                  // (the reference clamps the ratio; den > 0)
                if (num >= A.qos_s_hi * den) return true;
                if (num < A.qos_s_lo * den) return false;
                const double y = 1.0 + fmax(0.0, num / den);
                return log2(y) >= P.qos_R_min;
            };
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
                    okq = rate_ok(pf * gf, pn * gf + P.noise_power + 1e-12) && rate_ok(pn * gn, P.noise_power + 1e-12);
                }
                s_qos[lane + t * kWave] = okq;
            }
        }
        RISVEC_TICK(0)                                 // loads, history replay, QoS
        // ================= solve (TRAIN:1419-1524) ==================================================
        const int target = max(1, P.min_pair_target);
        double accept_q = P.mwm_accept_quantile;
        int K_back = A.K_back;
        double tau_b = A.tau_back[env];
        unsigned busy = 0;                             // wave-uniform: users already paired
        unsigned long long mate = 0;                   // 4 bits per user, valid where busy
        int rounds = 0, npairs = 0, K_last = 0;
        bool deferred = false;
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
            Sorted<EPL> R;
            int mine = 0;
            double Sv[EPL];                            // this lane's scores
            if (lane < kNV) s_adj[lane] = 0;           // targets of the graph-gathering atomics below
            if (lane == kNV) s_live[0] = 0;
#pragma unroll
            for (int t = 0; t < EPL; ++t) {
                double Rv = kInf;                      // finite <=> (feasible > 0) & isfinite(S); +inf ranks after everything
                Sv[t] = -kInf;
                if (ein[t]) {
                    const int idx = lane + t * kWave, i = ei[t], j = ej[t];
                    const double gap = fabs(s_g[i] - s_g[j]);
                    const bool abs_ok = !any_ok || s_g[i] >= P.abs_gain_min_db || s_g[j] >= P.abs_gain_min_db;
                    const float hterm = P.score_w_history * s_hist[idx];       // float32 product
                    double sv = P.score_w_delta_db * gap + (double)hterm;
                    if (!(s_feas[idx] && abs_ok)) sv = -kInf;
                    if (P.qos_enable && !s_qos[idx] && finite(sv)) sv = sv - P.qos_soft_penalty;
                    if (i == j) sv = -kInf;
                    s_S[idx] = sv;
                    Sv[t] = sv;
                    if (finite(sv)) { Rv = sv; ++mine; }
                }
                R.v[t] = Rv;
            }
            __syncthreads();
            sort_wave(R, lane);
            R.cnt = wave_sum(mine);
            busy = 0; mate = 0; npairs = 0; K_last = 0;
            RISVEC_TICK(1)                             // scores, ranks
            if (R.cnt > 0) {
                // ---- primary matching (TRAIN:326-398) -----------------------------------------------
                const double q = fmin(fmax(accept_q, 0.0), 1.0);
                const double thr = quantile_linear(R, 1.0 - q);
                // W = S where it reaches the threshold, -inf elsewhere; users without an edge drop out (singles allowed);
                // the rest, in increasing order, are the matcher's users 0..K-1
                {
                    // every lane knows which of its own entries are edges (i < j, S >= thr): the users with an edge, the
                    // compressed weights and the adjacency bits are gathered with LDS atomics instead of per-user scans
                    unsigned mine_users = 0;
                    bool edge[EPL];
#pragma unroll
                    for (int t = 0; t < EPL; ++t) {
                        edge[t] = ein[t] && ei[t] < ej[t] && finite(Sv[t]) && Sv[t] >= thr;
                        if (edge[t]) mine_users |= (1u << ei[t]) | (1u << ej[t]);
                    }
                    if (mine_users) atomicOr(&s_live[0], mine_users);
                    __syncthreads();
                    const unsigned live = singles ? s_live[0] : (N >= 32 ? ~0u : (1u << N) - 1u);
                    const int K = __popc(live);
                    K_last = K;
                    if (K > 0) {
#pragma unroll
                        for (int t = 0; t < EPL; ++t) {
                            if (!edge[t]) continue;
                            const int a = __popc(live & ((1u << ei[t]) - 1u)), b = __popc(live & ((1u << ej[t]) - 1u));
                            s_w[a * kNV + b] = Sv[t];
                            atomicOr(&s_adj[a], 1 << b);
                        }
                        __syncthreads();
                        RISVEC_TICK(2)                 // threshold, matchable users, compressed weights
                        if constexpr (!BIG) {          // up to 8 users: the table indexed by the mask
                            const PlainTab MT{s_dp, s_arg, (1 << K) - 1};
                            solve_plain(MT, s_w, s_adj, singles, K, lane);
                            walk_choices(MT, live, busy, mate, npairs);
                        } else {
                            // layer x (lane x): its frontier, how many of it can be taken, where its states start
                            unsigned meta = 0, adj_mine = 0;
                            int size = 0;
                            unsigned long long pw = 0;
                            if (lane < K) {
                                const unsigned above = ~((2u << lane) - 1u);
                                unsigned acc = 0;
                                int cap = 0;
                                for (int xp = 0; xp < lane; ++xp) {
                                    const unsigned ax = (unsigned)s_adj[xp];
                                    acc |= ax;
                                    cap += (ax & above) ? 1 : 0;
                                }
                                adj_mine = (unsigned)s_adj[lane];
                                const unsigned F = acc & above;
                                const int nf = __popc(F);
                                cap = min(cap, nf);
                                const bool direct = nf <= 6;           // 2^6: one pass of the wavefront either way
                                size = direct ? 1 << nf : colex->sizeoff[nf * (kBinW + 1) + cap + 1];
                                meta = F | ((unsigned)nf << 16) | ((unsigned)cap << 20) | ((direct ? 1u : 0u) << 24);
                                int r = 0;
                                for (unsigned m = F; m; m &= m - 1, ++r) pw |= (unsigned long long)r << (4 * (__ffs(m) - 1));
                            }
                            int base = size;                           // exclusive prefix sum over the 16 lanes that matter
    #pragma unroll
                            for (int o = 1; o <= kNV; o <<= 1) {
                                const int up = __shfl_up(base, o, kWave);
                                if (lane >= o) base += up;
                            }
                            base -= size;
                            if (lane < K) s_layer[lane] = Layer{pw, base, meta};
                            const int total = __shfl(base, K, kWave);  // lane K: everything below it
                            __syncthreads();
                            RISVEC_TICK(3)                 // frontiers
                            const int full = (1 << K) - 1;
                            if (MODE == 1 || total <= S::DP) {
                                if constexpr (STAMP) t_acc[10] += 1;
                                const FrontTab MT{s_dp, s_arg, s_layer, colex, full, total - 1};
                                solve_frontier<MODE == 1>(MT, s_w, meta, base, adj_mine, singles, K, lane);
                                RISVEC_TICK(4)             // table
                                walk_choices(MT, live, busy, mate, npairs);
                                RISVEC_TICK(5)             // walk
                            } else {                       // too large for this launch's table: leave the env to the second one
                                deferred = true;
                            }
                        }
                    }
                }
                RISVEC_TICK(2)
                // ---- greedy completion (TRAIN:276-324) ----------------------------------------------
                if (!deferred && npairs < target) {
                    const double thr2 = quantile_linear(R, P.completion_min_quantile);
                    while (npairs < target) {
                        double bs = -kInf;
                        int bi = -1;
                        // (unrolled by hand: `#pragma unroll` is refused for this loop in this nest, and a rolled loop
                        // indexes ei / ej dynamically, which moves them -- for the whole kernel -- into scratch memory)
                        const auto scan = [&](int t) {
                            if (!ein[t]) return;
                            const int idx = lane + t * kWave, i = ei[t], j = ej[t];
                            const double s = s_S[idx];
                            if (i < j && finite(s) && s >= thr2 && !((busy >> i) & 1) && !((busy >> j) & 1))
                                if (bi < 0 || s > bs || (s == bs && idx > bi)) { bs = s; bi = idx; }
                        };
                        scan(0);
                        if constexpr (EPL == 4) { scan(1); scan(2); scan(3); }
                        static_assert(EPL == 1 || EPL == 4, "scan() calls above");
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
            RISVEC_TICK(6)                             // completion
            if (deferred) break;
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
            RISVEC_TICK(7)                             // mask relaxation
        }
        __syncthreads();
        if (deferred) {                                // nothing of this env has been written yet
            if (lane == 0) {
                const int k = atomicAdd(&deferred_list->count, 1);
                if (k < A.ns.n_envs) deferred_list->env[k] = env;      // (always, on a list that started empty)
            }
            return;
        }
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
        RISVEC_TICK(8)                                 // stores
    };

    if constexpr (MODE == 0) {
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
    // Only the (rare) wavefronts that solve anything need the index arithmetic below; the opaque move keeps the
    // compiler from hoisting it into a prologue every frozen-step wavefront would then pay for.
    int tl = lane;
    asm volatile("" : "+v"(tl));
#pragma unroll
    for (int t = 0; t < EPL; ++t) {
        const int idx = tl + t * kWave;
        ein[t] = idx < NN;
        ei[t] = ein[t] ? idx / N : 0;
        ej[t] = ein[t] ? idx % N : 0;
    }
    }   // tables
    for (; todo; todo &= todo - 1) {
        const int slot = __ffs(todo) - 1;
        solve_env(e0 + slot, __shfl(my_flags, slot, kWave));
    }
    }   // 8-env groups
    } else {
        // second launch: the envs the first one left (their bookkeeping is done, nothing else of theirs was touched)
        const int n_left = min(deferred_list->count, A.ns.n_envs);
        if (n_left <= 0) return;                       // nearly every call: nothing was left
        {
            const int* src = reinterpret_cast<const int*>(&kColex);
            int* dst = reinterpret_cast<int*>(&s_colex);
            for (int i = lane; i < (int)(sizeof(ColexTab) / 4); i += kWave) dst[i] = src[i];
#pragma unroll
            for (int t = 0; t < EPL; ++t) {
                const int idx = lane + t * kWave;
                ein[t] = idx < NN;
                ei[t] = ein[t] ? idx / N : 0;
                ej[t] = ein[t] ? idx % N : 0;
            }
            for (int k = blockIdx.x; k < n_left; k += gridDim.x) {
                const int env = deferred_list->env[k];
                solve_env(env, (int)A.ns.flags[env]);
            }
        }
        // the last block to get here empties the list for the next call (every block has read `count` by then)
        __syncthreads();
        if (lane == 0) {
            __threadfence();
            if (atomicAdd(&deferred_list->done, 1) == (int)gridDim.x - 1) {
                deferred_list->count = 0;
                deferred_list->done = 0;
            }
        }
    }
#undef RISVEC_TICK
    if constexpr (STAMP) {
        long long* out = A.stamps + ((size_t)MODE * kDeferredGridMax * 64 + blockIdx.x) * kStamps;
        if (lane == 0 && (MODE == 1 || blockIdx.x < kDeferredGridMax * 64))
            for (int i = 0; i < kStamps; ++i) out[i] = t_acc[i];
    }
}

// tau = quantile q of |g_strong - g_weak| (TRAIN:842-855) and the feasibility mask (TRAIN:134-156).
// The quantile is over strong x weak user pairs only -- at most 8 x 8 = 64 values -- so they are packed (by the
// users' ranks, which are a permutation) into one value per lane before ranking: 64 x 64 compares instead of the
// 256 x 256 the full matrix cost at 16 users.
template <int NMAX>
__global__ void __launch_bounds__(kWave)
k_noma_mask(RisVecNomaState ns, const float* gain, const double* gdb15, double q_now, int K_now) {
    constexpr int NNM = NMAX * NMAX;
    __shared__ double s_g[kNV], s_d[NNM], s_R[kWave];
    __shared__ int s_rk[kNV];
    __shared__ uint8_t s_m[NNM], s_keep[NNM];
    const int lane = threadIdx.x;
    const int N = ns.n_veh, NN = N * N;
    const int n_weak = N / 2, n_pairs = (N - n_weak) * n_weak;
    for (int env = blockIdx.x; env < ns.n_envs; env += gridDim.x) {
        __syncthreads();
        if (lane < N)
            s_g[lane] = gdb15 ? gdb15[(long long)env * N + lane]
                              : 10.0 * log10(fmax((double)gain[(long long)env * N + lane], 1e-15));
        __syncthreads();
        // weak half = the n/2 smallest (argsort, equal keys by index); diffs over strong x weak
        if (lane < N) {
            int r = 0;
            for (int k = 0; k < N; ++k) r += (s_g[k] < s_g[lane] || (s_g[k] == s_g[lane] && k < lane)) ? 1 : 0;
            s_rk[lane] = r;
        }
        __syncthreads();
        for (int idx = lane; idx < NN; idx += kWave) {
            const int i = idx / N, j = idx % N;
            const int ri = s_rk[i], rj = s_rk[j];
            const double dgap = fabs(s_g[i] - s_g[j]);
            s_d[idx] = dgap;
            if (ri >= n_weak && rj < n_weak) s_R[(ri - n_weak) * n_weak + rj] = dgap;   // i strong, j weak
        }
        __syncthreads();
        double tau = 0.0;
        if (N >= 2) {
            Ranks<1> R;
            rank_entries<1>(s_R, n_pairs, lane, R);
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

// Blocks of k_noma_group's first launch (a wavefront each): one per EnvsPerWave envs, grid-stride beyond 2^20.
long long noma_group_blocks(int n_envs, int n_veh) {
    const int epw = n_veh <= 8 ? EnvsPerWave<8>::value : EnvsPerWave<16>::value;
    const long long waves = ((long long)n_envs + epw - 1) / epw;
    return waves < 1 ? 1 : (waves > (1 << 20) ? (1 << 20) : waves);
}
long long noma_stamp_bytes() {
#ifdef RISVEC_DIAG
    return (long long)(kDeferredGridMax * 64 + kDeferredGridMax) * kStamps * 8;
#else
    return 0;
#endif
}
// Up to 8 vehicles: none.  Beyond: the list of envs the first launch leaves to the second (16 B header + one int per env).
long long noma_scratch_bytes(int n_envs, int n_veh) {
    if (n_veh <= 8) return 0;
    return ((16 + 4LL * n_envs + 255) / 256) * 256 + noma_stamp_bytes();
}

// Start of an episode: hist / streak / pending / flags (and the list header of the two-launch scheme) zeroed in ONE launch
// (they were four to five memsets: a launch each, ~10 us per episode in a rollout loop).
__global__ void __launch_bounds__(kBlock)
k_noma_begin(RisVecNomaState ns, int n_scratch_words) {
    const long long E = ns.n_envs, N = ns.n_veh;
    const long long n_hist = E * N * N, n_streak = E * N;
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n_hist; i += stride) {
        ns.hist[i] = 0.0f;
        if (i < n_streak) ns.streak[i] = 0;
        if (i < E) { ns.pending[i] = 0; ns.flags[i] = 0; }
        if (i < n_scratch_words) static_cast<int*>(ns.scratch)[i] = 0;
    }
}

hipError_t launch_noma_begin_episode(const RisVecNomaState& ns, hipStream_t st) {
    const long long n_hist = (long long)ns.n_envs * ns.n_veh * ns.n_veh;       // the longest of the arrays (N >= 1)
    const int n_scratch_words = (ns.scratch && ns.scratch_bytes >= (long long)sizeof(Deferred)) ? (int)(sizeof(Deferred) / 4) : 0;
    long long blocks = (n_hist + kBlock - 1) / kBlock;
    if (blocks < 1) blocks = 1;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_noma_begin, dim3((unsigned)blocks), dim3(kBlock), 0, st, ns, n_scratch_words);
    return hipGetLastError();
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
    // TRAIN:870-871's log2(1 + sinr) >= R_min, see rate_ok: sinr >= s_hi certainly passes, sinr < s_lo certainly fails
    const double y = exp2(p.qos_R_min);
    const double s_hi = y * (1.0 + 1e-9) - 1.0 + 1e-9, s_lo = y * (1.0 - 1e-9) - 1.0 - 1e-9;
    long long* stamps = nullptr;
    if (noma_stamp_bytes() > 0 && ns.scratch)
        stamps = reinterpret_cast<long long*>(static_cast<char*>(ns.scratch) + noma_scratch_bytes(ns.n_envs, ns.n_veh) - noma_stamp_bytes());
    NomaArgs a{ns, p, gain, gdb12, p01, p01_raw, use_mask, K_back, tau_back, prev_global, prev_stride, i_step,
               u_unstick, seed, counter, info_out, ns.scratch, stamps, s_lo, s_hi};
    const dim3 grid((unsigned)noma_group_blocks(ns.n_envs, ns.n_veh));
    if (ns.n_veh <= 8) {
        hipLaunchKernelGGL((k_noma_group<8>), grid, dim3(kWave), 0, st, a);
        return hipGetLastError();
    }
    if (!ns.scratch || ns.scratch_bytes < noma_scratch_bytes(ns.n_envs, ns.n_veh)) return hipErrorInvalidValue;
    // second launch: one wavefront per env the first one left (none on most steps: its blocks read the count and leave)
    const long long want = ((long long)ns.n_envs + 31) / 32;
    const dim3 grid2((unsigned)(want < kDeferredGridMax ? (want < 1 ? 1 : want) : kDeferredGridMax));
#ifdef RISVEC_DIAG
    static const char* want_stamps = std::getenv("RISVEC_NOMA_STAMPS");
    if (want_stamps) {
        hipLaunchKernelGGL((k_noma_group<16, 0, true>), grid, dim3(kWave), 0, st, a);
        hipLaunchKernelGGL((k_noma_group<16, 1, true>), grid2, dim3(kWave), 0, st, a);
        return hipGetLastError();
    }
#endif
    hipLaunchKernelGGL((k_noma_group<16, 0>), grid, dim3(kWave), 0, st, a);
    hipLaunchKernelGGL((k_noma_group<16, 1>), grid2, dim3(kWave), 0, st, a);
    return hipGetLastError();
}

}  // namespace risvec
