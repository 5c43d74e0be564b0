// step() for one (env, vehicle) lane and its helpers, shared by k_step.hip (generic
// shapes) and k_step_pipe.hip (software-pipelined kernels for compile-time shapes).
//
// Reference: Simulation-MARL-BCD/Environment.py (ENV): compute_data_rate ENV:331-372,
// step ENV:547-731; observation marl_train_bcd.py:819-827, action map :1601-1608.
#pragma once

#include <type_traits>

#include "risvec_launch.hpp"

namespace risvec {

// Every RisVecParams field step() reads, as operands for RISVEC_ARGS_IN_ONE_TRIP.  RisVecParams sits at the front of
// the kernel-argument segment and these fields span three of its 64-byte lines; left alone the compiler loads each
// one right before its first use inside step(), and the first touch of a line is a cold scalar-cache miss (~0.3 us)
// on the critical path of a kernel that lives 4-6 us (seen in the ISA of k_step_fused_lat<8,36,2>: `s_load_dword
// .. 0x54` / `.. 0x68` each followed by `s_waitcnt lgkmcnt(0)` in the middle of the arithmetic).  Named here they
// travel with the pointers, in the one round trip every wavefront has to wait for anyway.
#define RISVEC_STEP_PARAMS(P)                                                                                       \
    "s"((P).bandwidth_mhz), "s"((P).noise_power), "s"((P).p_max), "s"((P).power_scale), "s"((P).qos_enable),          \
    "s"((P).r_min_bpshz), "s"((P).d_max_s), "s"((P).qos_penalty), "s"((P).time_fast), "s"((P).k_cpu),                 \
    "s"((P).f_local_max), "s"((P).f_edge_max), "s"((P).cycles_per_bit), "s"((P).cpu_share_floor), "s"((P).w_d),       \
    "s"((P).w_e), "s"((P).reward_clip), "s"((P).poisson_cdf[0]), "s"((P).poisson_cdf[1]), "s"((P).poisson_cdf[2]),    \
    "s"((P).poisson_cdf[3]), "s"((P).poisson_cdf[4]), "s"((P).poisson_cdf[5]), "s"((P).poisson_cdf[6]),               \
    "s"((P).poisson_cdf[7])

// The rollout's transition store, fused into the step (round 3; marl_train_bcd.py:1776-1799, buffer.py:16-25): the
// kernel that has the new observation, the rewards and the raw policy output of (env, vehicle) in registers also
// writes that agent's slice of the env's replay row -- what k_replay_store did in a second launch after re-reading
// obs / reward / metrics from HBM.  Row layout of buffer.py's seven arrays with input_shape = 5, n_actions = V + 2.
struct StepRing {
    float* state_memory;            // [mem_size, 5V]       the observation BEFORE this step (state.obs as the kernel finds it)
    float* action_memory;           // [mem_size, V(V+2)]   per agent [probs_i with zero diagonal | raw power_i]
    float* reward_global_memory;    // [mem_size]
    float* reward_local_memory;     // [mem_size, V]
    float* new_state_memory;        // [mem_size, 5V]
    uint8_t* terminal_memory;       // [mem_size]
    float* mask_memory;             // [mem_size, V V]
    const float* probs;             // [E, V, V] partner probabilities of the policy
    const uint8_t* mask;            // [E, V, V] NOMA mask bytes, or nullptr = all ones (marl_train_bcd.py:1786-1787)
    long long head, mem_size;       // ring row of env 0 (mem_cntr % mem_size), rows in the ring
    int done;                       // terminal flag of this step's transitions
};

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
    // theta by index (RISVEC_STEP_THETA_BY_INDEX): the BCD sweep's candidate indices, [E][theta_k_stride] bytes
    // (state.theta_idx); nullptr = read the complex64 theta
    const uint8_t* theta_k;
    int theta_k_stride;
    int ping;                       // non-temporal kernels: walk the envs in reverse this launch (step parity)
    StepRing ring;                  // used by the RING instances of the kernels only
};

// ---------------------------------------------------------------------------
// cross-lane exchange with a partner lane "O away".  O = 1, 2 are quad permutes,
// O = 4 / 8 use the DPP mirrors (partner = lane^7 / lane^15: it differs in bit 2 / bit 3,
// which is all a butterfly needs when steps run in monotone order), O = 16 is a
// ds_swizzle inside 32 lanes, O = 32 a bpermute.  DPP forms are plain VALU: no LDS
// crossbar round trip.
// ---------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}

template <int O>
__device__ __forceinline__ float xchg(float x) {
    if constexpr (O == 1) return dpp_mov<0xB1>(x);          // quad_perm [1,0,3,2]
    else if constexpr (O == 2) return dpp_mov<0x4E>(x);     // quad_perm [2,3,0,1]
    else if constexpr (O == 4) return dpp_mov<0x141>(x);    // row_half_mirror
    else if constexpr (O == 8) return dpp_mov<0x140>(x);    // row_mirror
    else if constexpr (O == 16)
        return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, x), 0x401F));
    else return __shfl_xor(x, 32, kWave);
}

// float64 exchange: two 32-bit exchanges
template <int O>
__device__ __forceinline__ double xchg(double x) {
    const float lo = xchg<O>(__builtin_bit_cast(float, __double2loint(x)));
    const float hi = xchg<O>(__builtin_bit_cast(float, __double2hiint(x)));
    return __hiloint2double(__builtin_bit_cast(int, hi), __builtin_bit_cast(int, lo));
}

// all-reduce (sum) over aligned groups of W lanes: a butterfly from the LARGEST distance down, i.e. the same
// pairings in the same order as the transposing reduction `treduce` below -- so a value summed by either comes out
// bit-identical (every lane evaluates the same tree: level k groups the lanes into the same partition whichever lane
// one starts from), which lets the latency-shaped kernels use `treduce` for the per-env metrics and still match the
// software pipeline to the last bit.
template <int W>
__device__ __forceinline__ double gsum(double x) {
    if constexpr (W >= 64) x += xchg<32>(x);
    if constexpr (W >= 32) x += xchg<16>(x);
    if constexpr (W >= 16) x += xchg<8>(x);
    if constexpr (W >= 8) x += xchg<4>(x);
    if constexpr (W >= 4) x += xchg<2>(x);
    if constexpr (W >= 2) x += xchg<1>(x);
    return x;
}

template <int W>
__device__ __forceinline__ float gsum(float x) {
    if constexpr (W >= 64) x += xchg<32>(x);
    if constexpr (W >= 32) x += xchg<16>(x);
    if constexpr (W >= 16) x += xchg<8>(x);
    if constexpr (W >= 8) x += xchg<4>(x);
    if constexpr (W >= 4) x += xchg<2>(x);
    if constexpr (W >= 2) x += xchg<1>(x);
    return x;
}

// transposing butterfly over a G-lane group: K values in, after log2(K) halving steps one
// value per lane, then plain all-reduce steps down to distance 1.
template <int K, int O, int N>
__device__ __forceinline__ void treduce(float (&val)[N], int gl) {
    if constexpr (O >= 1) {
        if constexpr (K > 1) {
            const bool hi = (gl & O) != 0;
#pragma unroll
            for (int j = 0; j < K / 2; ++j) {
                const float send = hi ? val[j] : val[j + K / 2];
                const float keep = hi ? val[j + K / 2] : val[j];
                val[j] = keep + xchg<O>(send);
            }
            treduce<K / 2, O / 2, N>(val, gl);
        } else {
            val[0] += xchg<O>(val[0]);
            treduce<1, O / 2, N>(val, gl);
        }
    }
}

// a / b as a * rcp(b): v_rcp_f32 is 1 ulp, the product adds 0.5 -> <= 1.5 ulp (1e-7), two
// instructions instead of the ~11 of an IEEE-rounded division.  Every denominator on this
// path is a normal float well inside rcp's range (noise power ~1e-14 ... cycles ~1e9).
__device__ __forceinline__ float fdiv(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }

// log2(1 + x) for x >= 0, relative error < 1e-6: below 1/16 the 1+x rounding would eat the
// result, so a 5-term log1p series is used there; above it v_log_f32(1 + x).
__device__ __forceinline__ float log2_1p(float x) {
    const float big = __builtin_amdgcn_logf(1.0f + x);
    const float ser = x * fmaf(x, fmaf(x, fmaf(x, fmaf(x, 0.2f, -0.25f), 0.33333334f), -0.5f), 1.f);
    return x < 0.0625f ? ser * 1.4426950408889634f : big;
}

// inputs of step() for one lane, loadable ahead of the compute
struct StepIn {
    float a0, a1, B, Q0, pl;
    int part, G;
    int arr_in;              // injected arrivals (ignored when the draw is Philox)
};

// Where a lane's injected arrival count lives -- ALWAYS a valid address: without injected arrivals the lane re-reads
// its `partner` word (same cache line as the load next to it, value ignored).  A load under `if (A.arrivals)` is a
// branch around a request, and the compiler can only join the two paths with `s_waitcnt vmcnt(0)`: a full drain of
// every h_r request in flight, in front of the arithmetic that was supposed to hide under them.
__device__ __forceinline__ const int32_t* arrivals_src(const StepArgs& A) { return A.arrivals ? A.arrivals : A.partner; }

__device__ __forceinline__ StepIn load_step_in(const Dims& d, const StepArgs& A, int e, int v, bool active) {
    StepIn in{0.f, 0.f, 0.f, 0.f, 0.f, RISVEC_PARTNER_NONE, 1, 0};
    if (active) {
        const int V = d.V;
        const long long idx = (long long)e * V + v;
        if (A.flags & RISVEC_STEP_POLICY_ACTION) {
            const float2 pa = *reinterpret_cast<const float2*>(A.action + idx * 2);
            in.a0 = pa.x;
            in.a1 = pa.y;
        } else {
            in.a0 = A.action[(long long)e * 2 * V + v];
            in.a1 = A.action[(long long)e * 2 * V + V + v];
        }
        in.B = A.data_buf[idx];
        in.part = A.partner[idx];
        in.arr_in = arrivals_src(A)[idx];
        in.G = A.n_groups[e];
        in.Q0 = A.mec_q[e];
        in.pl = A.pl ? A.pl[idx] : 0.f;
    }
    return in;
}

// compute_data_rate (ENV:331-372) for one lane; all lanes of the VP-group must call.
// near = u1 if gain1 > gain2 else u2 (ENV:355-360): a vehicle listed second is "near" on ties.
template <int VP>
__device__ __forceinline__ float noma_rate(const RisVecParams& P, float pw0, float gain, int part, int G) {
#pragma clang fp contract(off)     // fused products are written out (fmaf): no rounding may depend on the calling kernel
    const int lane = threadIdx.x & (kWave - 1);
    const int base = lane & ~(VP - 1);
    const bool pair = part >= 0, single = part == RISVEC_PARTNER_SINGLE;
    const bool second = part >= RISVEC_PARTNER_SECOND;
    const int src = base + (pair ? (part & (VP - 1)) : (lane - base));
    const float g_p = __shfl(gain, src, kWave);
    const float pw_p = __shfl(pw0, src, kWave);
    const bool near = second ? !(g_p > gain) : (gain > g_p);
    const float sig = pw0 * gain;                                            // ENV:347, 362, 367
    const float den = (pair && !near) ? fmaf(pw_p, gain, P.noise_power) : P.noise_power;   // ENV:363-364
    const float sinr = fdiv(sig, den);
    const float frac = __builtin_amdgcn_rcpf((float)max(1, G));              // ENV:341-342
    const float rate = frac * log2_1p(sinr);
    return (pair || single) ? rate : 0.f;
}

// gain from the reduced cascade sum: | sqrt(pl) img + h_d |^2  (ENV:270-272; h_d = 0 there)
__device__ __forceinline__ float gain_from_img(float2 img, float pl, const float* h_d, long long idx) {
    if (h_d) {
        const float a = sqrtf(pl);
        const float2 hd = *reinterpret_cast<const float2*>(h_d + idx * 2);
        const float re = fmaf(a, img.x, hd.x), im = fmaf(a, img.y, hd.y);
        return fmaf(re, re, im * im);
    }
    return pl * fmaf(img.x, img.x, img.y * img.y);     // explicit: the same rounding in every kernel
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
    acc.x = gsum<G>(acc.x);
    acc.y = gsum<G>(acc.y);
    return acc;
}

// lanes per (env, vehicle) row: enough to cover M/VEC elements in one pass when
// possible, but never more vehicles per pass than an env has (keeps lanes busy).
inline int pick_group(int M, int vec, int VP) {
    int g = pow2_ceil((M + vec - 1) / vec);
    if (g > kWave) g = kWave;
    const int gmin = kWave / VP;             // >= this => vehicles per pass <= VP
    if (g < gmin) g = gmin;
    if (g < 8) g = 8;
    return g;
}

// store of an output nobody reads again before the next kernel (experiment switch: non-temporal hint)
template <class T>
__device__ __forceinline__ void st_out(T* p, T v) {
#ifdef RISVEC_NT_STORES
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

// Per-step extras of the multi-step launch (risvec_step_fused_multi): where this step's trajectory
// record goes (pointers already offset to the step's slice; nullptr = not recorded) and whether the
// env's own state / output tensors are written (only the last step of a launch needs to).
struct StepTraj {
    float* reward;      // [E,V]
    float* obs;         // [E,V,5]
    float* metrics;     // [E,16]
    bool store_state;
};

// new queue state a step leaves behind (kept in registers across the steps of one launch)
struct StepCarry {
    float B, Q;
};

// ---------------------------------------------------------------------------
// step() for one (env, vehicle) lane, in two halves (round 3):
//   step_pre   everything that does NOT depend on the channel gain -- action map, power projection, CPU share,
//              local processing, local energy, the Philox + Poisson arrival draw, over_power.  About a third of
//              step()'s instructions; the latency-shaped kernels run it while their h_r / theta requests are in
//              flight (a wavefront there is a chain  issue -> memory round trip -> reduce -> step() -> stores).
//   step_tail  the rest, from the NOMA rate on.  Called by ALL 64 lanes (cross-lane ops inside); `active` masks
//              lanes beyond V or E.
// step_core = step_pre + step_tail back to back: the same operations on the same values whichever kernel calls
// them in whichever order, so the kernels stay bit-identical (tests/test_entry_points_hip.py).
// TRAJ = false is the single-step form (every output goes to the env's tensors); TRAJ = true additionally honours `tj`.
// TM = true: the per-env metrics through one transposing reduction (fewer instructions: the choice of the
// latency-shaped kernels); TM = false: one butterfly per metric, lane 0 stores four float4 (the software pipeline:
// its 16-byte stores are a little kinder to a kernel that is busy streaming h_r).  Same values bit for bit.
// ---------------------------------------------------------------------------
// per-lane inputs of the fused transition store, requested with the step inputs
template <int VP>
struct RingIn {
    float so[5];                    // the observation this step replaces
    float prow[VP];                 // probs[e, v, :]
    unsigned mk[VP / 4];            // mask[e, v, :] bytes
};

template <int VP>
__device__ __forceinline__ RingIn<VP> load_ring_in(const Dims& d, const StepArgs& A, int e, int v, bool active) {
    RingIn<VP> r;
    const long long idx = active ? (long long)e * VP + v : 0;       // inactive lanes re-read (env 0, vehicle 0): no branch
    const float* o = A.obs + idx * 5;
#pragma unroll
    for (int k = 0; k < 5; ++k) r.so[k] = o[k];
    const float4* p4 = reinterpret_cast<const float4*>(A.ring.probs + idx * VP);
#pragma unroll
    for (int k = 0; k < VP / 4; ++k) {
        const float4 x = p4[k];
        r.prow[4 * k] = x.x; r.prow[4 * k + 1] = x.y; r.prow[4 * k + 2] = x.z; r.prow[4 * k + 3] = x.w;
    }
    // all-ones without a mask: the address is selected, not the load (the probs row stands in; its value is ignored)
    const unsigned* m4 = A.ring.mask ? reinterpret_cast<const unsigned*>(A.ring.mask + idx * VP)
                                     : reinterpret_cast<const unsigned*>(A.ring.probs + idx * VP);
#pragma unroll
    for (int k = 0; k < VP / 4; ++k) r.mk[k] = m4[k];
    return r;
}

__device__ __forceinline__ void ring_st(float* p, float v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void ring_st2(float* p, float a, float b) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f t; t.x = a; t.y = b;
    __builtin_nontemporal_store(t, reinterpret_cast<v2f*>(p));
}
__device__ __forceinline__ void ring_st4(float* p, float a, float b, float c, float dd) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f t; t.x = a; t.y = b; t.z = c; t.w = dd;
    __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(p));
}

// This env's replay row (the ring is write-once data: non-temporal stores), written by the VP lanes of the env TOGETHER.
// A lane owns vehicle v's pieces of the row -- 5 + 5 floats of the two states, VP + 2 of the action, VP of the mask -- at a
// stride that is not a multiple of 16 bytes, so storing them straight from the lane is 4- and 8-byte stores 20 / 40 bytes
// apart: 18 store instructions per wavefront, each touching every line of the segment for a fifth of its bytes.  The row
// pieces go through the env's slice of LDS instead (written as owned, read back as consecutive 16-byte chunks): 10 store
// instructions, each a whole run of the row.  32 768 x 8: rollout step with the ring 41.6 -> 39.2 us (the same bytes
// written perfectly coalesced as a test gave the same 39.2: profiles/r03aw_ring_coalesced_ab.txt).  Same-wave LDS
// accesses execute in order and an env's lanes touch nothing but its slice: no barrier.
template <int VP>
__device__ __forceinline__ void ring_store(const StepArgs& A, int e, int v, const RingIn<VP>& R, const StepIn& in, float Bn,
                                           float data_t, float data_p, float rate, float rew) {
    static_assert(VP % 4 == 0, "rows are split into 16-byte chunks per lane");
    constexpr int SL = VP * (VP + 2);                          // floats of the longest row piece (the action)
    __shared__ float s_row[kBlock / kWave][(kWave / VP) * SL];
    float* sl = s_row[threadIdx.x / kWave] + ((threadIdx.x % kWave) / VP) * SL;   // this env's slice
    const StepRing& G = A.ring;
    long long row = G.head + e;
    row = row >= G.mem_size ? row - G.mem_size : row;
    // a row piece of 4 VP + VP floats (a state): 16 bytes + 4 bytes per lane
    const auto put_state = [&](float* dst, float x0, float x1, float x2, float x3, float x4) {
        float* mine = sl + 5 * v;
        mine[0] = x0; mine[1] = x1; mine[2] = x2; mine[3] = x3; mine[4] = x4;
        const float4 q = *reinterpret_cast<const float4*>(sl + 4 * v);
        const float tail = sl[4 * VP + v];
        ring_st4(dst + 4 * v, q.x, q.y, q.z, q.w);
        ring_st(dst + 4 * VP + v, tail);
    };
    put_state(G.state_memory + row * (5 * VP), R.so[0], R.so[1], R.so[2], R.so[3], R.so[4]);
    put_state(G.new_state_memory + row * (5 * VP), Bn * 0.1f, data_t * 0.1f, data_p * 0.1f, 0.f, rate * 0.05f);
    ring_st(G.reward_local_memory + row * VP + v, rew);
    {   // action: VP (VP + 2) floats = VP/4 runs of 4 VP + one of 2 VP
        float2* mine = reinterpret_cast<float2*>(sl + (VP + 2) * v);           // 8-byte aligned: VP + 2 is even
#pragma unroll
        for (int k = 0; k < VP; k += 2) mine[k / 2] = make_float2(k == v ? 0.f : R.prow[k], k + 1 == v ? 0.f : R.prow[k + 1]);
        mine[VP / 2] = make_float2(in.a0, in.a1);                              // the raw policy output (TRAIN:1777-1782)
        float* a = G.action_memory + row * (VP * (VP + 2));
#pragma unroll
        for (int p = 0; p < VP / 4; ++p) {
            const float4 q = *reinterpret_cast<const float4*>(sl + p * 4 * VP + 4 * v);
            ring_st4(a + p * 4 * VP + 4 * v, q.x, q.y, q.z, q.w);
        }
        const float2 t2 = *reinterpret_cast<const float2*>(sl + VP * VP + 2 * v);
        ring_st2(a + VP * VP + 2 * v, t2.x, t2.y);
    }
    {   // mask: VP VP floats = VP/4 runs of 4 VP
        const bool ones = G.mask == nullptr;
        float4* mine = reinterpret_cast<float4*>(sl + VP * v);
#pragma unroll
        for (int k = 0; k < VP / 4; ++k) {
            const unsigned w = R.mk[k];
            mine[k] = make_float4((ones || (w & 0xFFu)) ? 1.f : 0.f, (ones || (w & 0xFF00u)) ? 1.f : 0.f,
                                  (ones || (w & 0xFF0000u)) ? 1.f : 0.f, (ones || (w & 0xFF000000u)) ? 1.f : 0.f);
        }
        float* m = G.mask_memory + row * (VP * VP);
#pragma unroll
        for (int p = 0; p < VP / 4; ++p) {
            const float4 q = *reinterpret_cast<const float4*>(sl + p * 4 * VP + 4 * v);
            ring_st4(m + p * 4 * VP + 4 * v, q.x, q.y, q.z, q.w);
        }
    }
    if (v == 0) G.terminal_memory[row] = (uint8_t)(G.done ? 1 : 0);
}

struct StepPre {
    float pw0, pw1;          // transmit / (unused) second power, W                     ENV:555-561
    float f;                 // local CPU frequency                                      ENV:572-580
    float bc, cap, used;     // backlog cycles, cycle budget, cycles spent locally       ENV:585-592
    float data_p, rem;       // kbit processed locally, kbit left for offloading
    float E_loc;             // local energy                                             ENV:664
    float over_power;        //                                                          ENV:721
    float arr_kbit;          // this step's arrivals                                     ENV:717-719
};

// ARR_LATE (the T-step loop): the arrival draw is left to step_tail, at the place step() has it.  There the injected
// arrivals are loaded under `if (A.arrivals)`, whose join costs an `s_waitcnt vmcnt(0)`: late in the step that is free,
// at the top of the step it would wait for the next step's action prefetch every step (measured: +14 % per step).
template <bool ARR_LATE = false>
__device__ __forceinline__ StepPre step_pre(const Dims& d, const RisVecParams& P, const StepArgs& A, int e, int v,
                                            bool active, const StepIn& in) {
#pragma clang fp contract(off)     // see step_tail
    StepPre r;
    const float B = in.B;
    float a0 = in.a0, a1 = in.a1;

    // ENV:574-577
    float fl = P.cpu_share_floor;
    if (!isfinite(fl)) fl = 0.10f;
    fl = fmaxf(0.f, fminf(fl, 0.95f));
    if (A.flags & RISVEC_STEP_POLICY_ACTION) {
        // marl_train_bcd.py:1601-1608: [-1,1] -> [0,1], CPU share floored
        a0 = (fminf(fmaxf(a0, -0.999f), 0.999f) + 1.f) * 0.5f;
        a1 = fmaxf((fminf(fmaxf(a1, -0.999f), 0.999f) + 1.f) * 0.5f, fl);
    }

    // (1) power projection, ENV:555-561
    float c0 = fmaxf(a0, 0.f) * P.power_scale;
    float c1 = fmaxf(a1, 0.f) * P.power_scale;
    const float s = c0 + c1;
    if (s > 1.f) {
        const float inv = __builtin_amdgcn_rcpf(s + 1e-12f);
        c0 = c0 * inv;
        c1 = c1 * inv;
    }
    r.pw0 = c0 * P.p_max;
    r.pw1 = c1 * P.p_max;

    // (3) cpu share, ENV:572-580
    const float cpu = fmaxf(fminf(fmaxf(a1, 0.f), 1.f), fl);
    r.f = cpu * P.f_local_max;
    const float Cpb = P.cycles_per_bit, tf = P.time_fast;

    // (4) local processing, ENV:585-592
    // When the CPU can clear the whole backlog, data_p = bc / (Cpb*1000) equals B up to
    // rounding (1e-16 in the float64 reference).  In float32 that rounding (1e-7 B) would
    // leak into `rem`, `off` and t_tx = off / throughput, so the identity is used directly.
    r.bc = B * 1000.0f * Cpb;
    r.cap = r.f * tf;
    const bool clears = r.cap >= r.bc;
    r.used = clears ? r.bc : r.cap;
    r.data_p = clears ? B : fdiv(r.cap, Cpb * 1000.0f);
    r.rem = fmaxf(0.f, B - r.data_p);                                       // ENV:595

    r.E_loc = P.k_cpu * (r.f * r.f) * r.used;                                // ENV:664

    // (12) arrivals, ENV:717-719
    // The Philox draw is the production path and must be the FALL-THROUGH: a null-pointer test is predicted "unlikely",
    // which put the draw out of line -- two taken branches (two instruction-fetch restarts) per step for a lone wavefront.
    int arr = 0;
    if constexpr (!ARR_LATE) {
        if (__builtin_expect(A.arrivals == nullptr, 1)) {
            const uint4 x = philox4x32_10((uint32_t)(d.env_offset + e), (uint32_t)v, A.counter, kSiteArrivals, A.seed);
            arr = poisson_from_u(u01(x.x), P.poisson_cdf);
        } else {
            arr = in.arr_in;
        }
    }
    r.arr_kbit = (float)arr * tf * 1000.0f;

    r.over_power = fmaxf(0.f, (r.pw0 + r.pw1) - P.p_max);                     // ENV:721
    return r;
}

struct NoRing {};

template <int VP, bool TRAJ = false, bool TM = false, class RIN = NoRing>
__device__ __forceinline__ StepCarry step_tail(const Dims& d, const RisVecParams& P, const StepArgs& A,
                                               int e, int v, bool active, float gain, const StepIn& in,
                                               const StepPre& pre, const StepTraj* tj = nullptr, const RIN* rin = nullptr) {
    // No implicit contraction: whether `a * b + c` became one fused instruction used to depend on what else the
    // compiler saw around it (the early step_pre of the latency-shaped kernels vs the back-to-back form of the
    // software pipeline), and the kernels must agree to the last bit.  The fusions worth having are written out.
#pragma clang fp contract(off)
    constexpr bool RING = !std::is_same<RIN, NoRing>::value;
    const int V = d.V;
    const long long idx = (long long)e * V + v;
    const float eps = 1e-12f;
    const float B = in.B, Q0 = in.Q0;
    const int part = in.part, G = in.G;
    const float pw0 = pre.pw0, f = pre.f, bc = pre.bc, cap = pre.cap, used = pre.used, data_p = pre.data_p;
    const float Cpb = P.cycles_per_bit;

    // (2) rate, ENV:331-372
    const float rate = noma_rate<VP>(P, pw0, gain, part, G);
    const float tf = P.time_fast, bw = P.bandwidth_mhz;
    const float data_t = rate * tf * bw * 1000.0f;                          // ENV:570

    // (5) offload, ENV:595-601
    const float off = fminf(data_t, pre.rem);
    const float thr = rate * bw * 1000.0f;
    const float t_tx = fdiv(off, thr + 1e-12f);

    // (6) MEC queue, ENV:604-610
    const float ein = off * 1000.0f * Cpb;
    const float ein_sum = gsum<VP>(active ? ein : 0.f);
    float Q = Q0 + ein_sum;
    const float edge_cap = P.f_edge_max * tf;
    const float svc = fminf(edge_cap, Q);
    Q -= svc;

    // (7) backlog, ENV:617-618
    float Bn = fmaxf(0.f, B - (data_p + off));

    // (8) delays, ENV:622-633
    const float inv_fe = __builtin_amdgcn_rcpf(P.f_edge_max + eps);
    const float d_loc = fdiv(fmaxf(0.f, bc - ein), f + eps);
    const float share = fdiv(ein, ein_sum + eps);
    const float d_q = share * (Q0 * inv_fe);
    const float d_c = ein * inv_fe;
    const float delay = d_loc + t_tx + d_q + d_c;

    // (9) energy, ENV:659-666
    const float E_tx = pw0 * t_tx;
    const float E_loc = pre.E_loc;
    const float energy = E_tx + E_loc;

    // (10) QoS, ENV:669-677
    const bool viol = P.qos_enable && ((rate < P.r_min_bpshz) || (delay > P.d_max_s));
    const float pen = viol ? P.qos_penalty : 0.f;

    // (11) reward, ENV:696-703
    const float cost = fmaf(P.w_d, delay, P.w_e * energy);
    const float rew = fminf(fmaxf(-cost - pen, -P.reward_clip), P.reward_clip);

    // (12) arrivals, ENV:717-719 (drawn in step_pre, except in the T-step loop)
    if constexpr (TRAJ) {
        int arr = 0;
        if (A.arrivals) {
            if (active) arr = A.arrivals[idx];
        } else {
            const uint4 x = philox4x32_10((uint32_t)(d.env_offset + e), (uint32_t)v, A.counter, kSiteArrivals, A.seed);
            arr = poisson_from_u(u01(x.x), P.poisson_cdf);
        }
        Bn += (float)arr * tf * 1000.0f;
    } else {
        Bn += pre.arr_kbit;
    }

    // (13) ENV:721-729
    const float over_power = pre.over_power;
    const float inv_v = __builtin_amdgcn_rcpf((float)V);

    bool store_state = true;
    if constexpr (TRAJ) store_state = tj->store_state;
    if (active && store_state) {
        A.data_buf[idx] = Bn;
        st_out(A.rate + idx, rate);
        st_out(A.data_t + idx, data_t);
        st_out(A.data_p + idx, data_p);
        st_out(A.reward + idx, rew);
        st_out(A.over_power + idx, over_power);
        if (A.flags & RISVEC_STEP_OBS) {
            // marl_train_bcd.py:819-827 (element 3 = over_data/10 is always 0)
            float* o = A.obs + idx * 5;
            st_out(o + 0, Bn * 0.1f); st_out(o + 1, data_t * 0.1f); st_out(o + 2, data_p * 0.1f); st_out(o + 3, 0.f);
            st_out(o + 4, rate * 0.05f);
        }
        if (A.flags & RISVEC_STEP_POWER_W) {
            const float inv_tf = __builtin_amdgcn_rcpf(tf);
            A.power_w[(long long)e * 2 * V + v] = E_tx * inv_tf;
            A.power_w[(long long)e * 2 * V + V + v] = E_loc * inv_tf;
        }
        if (v == 0) A.mec_q[e] = Q;
        if constexpr (RING) ring_store<VP>(A, e, v, *rin, in, Bn, data_t, data_p, rate, rew);
    }
    if constexpr (TRAJ) {
        if (active) {
            if (tj->reward) tj->reward[idx] = rew;
            if (tj->obs) {
                float* o = tj->obs + idx * 5;
                o[0] = Bn * 0.1f; o[1] = data_t * 0.1f; o[2] = data_p * 0.1f; o[3] = 0.f; o[4] = rate * 0.05f;
            }
        }
    }

    if constexpr (TM && (VP == 4 || VP == 8 || VP == 16)) {
        if ((A.flags & RISVEC_STEP_METRICS) && V == VP) {
            // The 14 per-env scalars as ONE transposing reduction per VP values instead of one 3-/4-step butterfly
            // each (36 exchanges -> 14 at VP = 8): slot s is contributed by every lane of the env (the two
            // per-env values Q and the MEC utilisation by lane 0 only), lands on lane s % VP, which scales and
            // stores it -- metrics[e][s] leaves as one float per lane, 64 contiguous bytes per env.
            const float z = 0.f;
            const bool lead = active && v == 0;
            float c[16];
            c[0] = active ? rew : z;  c[1] = active ? off : z;  c[2] = active ? data_p : z;  c[3] = lead ? Q : z;
            c[4] = active ? B : z;    c[5] = active ? d_loc : z; c[6] = active ? d_q : z;    c[7] = active ? d_c : z;
            c[8] = active ? t_tx : z; c[9] = lead ? fdiv(svc, edge_cap + 1e-12f) : z;
            c[10] = active ? fdiv(used, cap + 1e-12f) : z;       c[11] = (active && viol) ? 1.f : z;
            c[12] = active ? delay : z; c[13] = active ? energy : z; c[14] = z; c[15] = z;
            bool store_state = true;
            if constexpr (TRAJ) store_state = tj->store_state;
#pragma unroll
            for (int p0 = 0; p0 < 16; p0 += VP) {
                float part[VP];
#pragma unroll
                for (int j = 0; j < VP; ++j) part[j] = c[p0 + j];
                treduce<VP, VP / 2, VP>(part, v);
                const int slot = p0 + v;
                const bool is_sum = slot == 1 || slot == 2 || slot == 3 || slot == 9;
                const float val = part[0] * (is_sum ? 1.f : inv_v);
                if (active) {
                    if constexpr (RING) {
                        if (slot == 0) {
                            long long row = A.ring.head + e;
                            row = row >= A.ring.mem_size ? row - A.ring.mem_size : row;
                            ring_st(A.ring.reward_global_memory + row, val);
                        }
                    }
                    if (store_state) A.metrics[(long long)e * RISVEC_METRICS + slot] = val;
                    if constexpr (TRAJ) {
                        if (tj->metrics) tj->metrics[(long long)e * RISVEC_METRICS + slot] = val;
                    }
                }
            }
            return StepCarry{Bn, Q};
        }
    }
    const float rew_sum = gsum<VP>(active ? rew : 0.f);
    if constexpr (RING) {
        if (active && v == 0) {
            long long row = A.ring.head + e;
            row = row >= A.ring.mem_size ? row - A.ring.mem_size : row;
            ring_st(A.ring.reward_global_memory + row, rew_sum * inv_v);
        }
    }
    if (A.flags & RISVEC_STEP_METRICS) {
        const float z = 0.f;
        const float s_off = gsum<VP>(active ? off : z);
        const float s_dp = gsum<VP>(active ? data_p : z);
        const float s_b = gsum<VP>(active ? B : z);
        const float s_dl = gsum<VP>(active ? d_loc : z);
        const float s_dq = gsum<VP>(active ? d_q : z);
        const float s_dc = gsum<VP>(active ? d_c : z);
        const float s_tx = gsum<VP>(active ? t_tx : z);
        const float s_ut = gsum<VP>(active ? fdiv(used, cap + 1e-12f) : z);
        const float s_vi = gsum<VP>(active && viol ? 1.f : z);
        const float s_de = gsum<VP>(active ? delay : z);
        const float s_en = gsum<VP>(active ? energy : z);
        if (active && v == 0) {
            const float4 m0 = make_float4(rew_sum * inv_v, s_off, s_dp, Q);
            const float4 m1 = make_float4(s_b * inv_v, s_dl * inv_v, s_dq * inv_v, s_dc * inv_v);
            const float4 m2 = make_float4(s_tx * inv_v, fdiv(svc, edge_cap + 1e-12f), s_ut * inv_v, s_vi * inv_v);
            const float4 m3 = make_float4(s_de * inv_v, s_en * inv_v, 0.f, 0.f);
            if (store_state) {
                float4* m = reinterpret_cast<float4*>(A.metrics + (long long)e * RISVEC_METRICS);
                m[0] = m0; m[1] = m1; m[2] = m2; m[3] = m3;
            }
            if constexpr (TRAJ) {
                if (tj->metrics) {
                    float4* m = reinterpret_cast<float4*>(tj->metrics + (long long)e * RISVEC_METRICS);
                    m[0] = m0; m[1] = m1; m[2] = m2; m[3] = m3;
                }
            }
        }
    } else if (active && v == 0) {
        if (store_state) A.metrics[(long long)e * RISVEC_METRICS] = rew_sum * inv_v;          // global_reward only
        if constexpr (TRAJ) {
            if (tj->metrics) tj->metrics[(long long)e * RISVEC_METRICS] = rew_sum * inv_v;
        }
    }
    return StepCarry{Bn, Q};
}

template <int VP, bool TRAJ = false, bool TM = false, class RIN = NoRing>
__device__ __forceinline__ StepCarry step_core(const Dims& d, const RisVecParams& P, const StepArgs& A,
                                               int e, int v, bool active, float gain, const StepIn& in,
                                               const StepTraj* tj = nullptr, const RIN* rin = nullptr) {
    const StepPre pre = step_pre<TRAJ>(d, P, A, e, v, active, in);
    return step_tail<VP, TRAJ, TM, RIN>(d, P, A, e, v, active, gain, in, pre, tj, rin);
}

inline StepArgs make_step_args(const RisVecState& s, const float* action, const int32_t* partner,
                               const int32_t* n_groups, const int32_t* arrivals, uint64_t seed,
                               uint32_t counter, uint32_t flags) {
    StepArgs a;
    a.action = action; a.partner = partner; a.n_groups = n_groups; a.arrivals = arrivals;
    a.pl = s.pl; a.h_r = s.h_r; a.theta = s.theta; a.b = s.b; a.h_d = s.h_d;
    a.gain = s.gain; a.data_buf = s.data_buf; a.mec_q = s.mec_q;
    a.rate = s.rate; a.data_t = s.data_t; a.data_p = s.data_p; a.reward = s.reward;
    a.over_power = s.over_power; a.obs = s.obs; a.metrics = s.metrics; a.power_w = s.power_w;
    a.seed = seed; a.counter = counter; a.flags = flags;
    a.theta_k = nullptr; a.theta_k_stride = 0; a.ping = 0;
    a.ring = StepRing{};
    return a;
}

// specialised software-pipelined fused kernels (k_step_pipe.hip); returns
// hipErrorNotSupported when the shape has no specialisation.
hipError_t launch_step_fused_pipe(const RisVecState& s, const RisVecParams& p, const StepArgs& a,
                                  hipStream_t st);
hipError_t launch_gain_pipe(const RisVecState& s, hipStream_t st);
hipError_t launch_step_fused_pipe_ring(const RisVecState& s, const RisVecParams& p, const StepArgs& a, hipStream_t st);
// latency-shaped single-group kernels for small batches and the multi-step launch (k_step_lat.hip)
hipError_t launch_step_fused_lat(const RisVecState& s, const RisVecParams& p, const StepArgs& a, hipStream_t st);
bool step_fused_lat_covers(int V, int M);
bool theta_by_index_supported(int V, int M);
hipError_t launch_step_fused_multi(const RisVecState& s, const RisVecParams& p, const StepArgs& a, int n_steps,
                                   const RisVecTraj* traj, hipStream_t st);
// n_steps consecutive steps on the CACHED gains in one launch, any shape (k_step.hip)
hipError_t launch_step_multi(const RisVecState& s, const RisVecParams& p, const StepArgs& a, int n_steps,
                             const RisVecTraj* traj, hipStream_t st);

// The step loop of a T-step launch (SURVEY 7 "launch latency"; the driver loop marl_train_bcd.py:1304-1611 between two
// channel refreshes, groups frozen as inside an episode), shared by the fused form (k_step_fused_lat<.., MULTI>) and the
// cached-gain form (k_step_multi): the wavefront keeps its envs' queues in registers and walks the steps.  `in` holds
// step 0's inputs; actions[t + 1] is in flight during step t.  Everything that moves from step to step ADVANCES by a
// stride (zero for an absent buffer: a null pointer stays null) instead of being recomputed as `p ? p + t * stride :
// nullptr` -- that was a branch around four scalar instructions per buffer per step, seven branches (and a
// kernel-argument reload) in a loop whose lone wavefront pays an instruction-fetch restart for each.
template <int VP>
__device__ __forceinline__ void multi_step_loop(const Dims& d, const RisVecParams& P, const StepArgs& A, const RisVecTraj& TJ,
                                                int e, int v, bool active, float g, StepIn in, int n_steps) {
    const int V = d.V;
    const long long ev = (long long)d.E * V, idx = (long long)e * V + v;
    const bool pol = (A.flags & RISVEC_STEP_POLICY_ACTION) != 0;
    StepArgs At = A;
    StepTraj tj;
    tj.reward = TJ.reward; tj.obs = TJ.obs; tj.metrics = TJ.metrics;
    const long long rw_stride = TJ.reward ? ev : 0, ob_stride = TJ.obs ? ev * 5 : 0;
    const long long mt_stride = TJ.metrics ? (long long)d.E * RISVEC_METRICS : 0;
    const long long ar_stride = A.arrivals ? ev : 0;
    // this lane's word(s) of action[t]: [T, E, V, 2] (policy layout) or [T, E, 2, V]
    // (inactive lanes re-read the action of (env 0, vehicle 0): an address select, not a branch around the two loads)
    const float* ap = A.action + (active ? (pol ? idx * 2 : (long long)e * 2 * V + v) : 0);
    const long long a1_off = pol ? 1 : V;
    // The last step is peeled: only it writes the env's own tensors, so inside the loop `store_state` is a constant
    // false -- no branch on it, and the eleven output pointers (spilled: the kernel is at the SGPR limit) are not
    // read back from their spill lanes once per step.
    tj.store_state = false;
#pragma unroll 1
    for (int t = 0; t + 1 < n_steps; ++t) {
        // next step's action: in flight during this step's arithmetic
        ap += 2 * ev;
        const float a0n = ap[0], a1n = ap[a1_off];
        const StepCarry c = step_core<VP, true, true>(d, P, At, e, v, active, g, in, &tj);
        At.counter += 1u;
        At.arrivals += ar_stride;
        tj.reward += rw_stride;
        tj.obs += ob_stride;
        tj.metrics += mt_stride;
        in.B = c.B;
        in.Q0 = c.Q;
        in.a0 = a0n;
        in.a1 = a1n;
    }
    tj.store_state = true;
    step_core<VP, true, true>(d, P, At, e, v, active, g, in, &tj);
}

}  // namespace risvec
