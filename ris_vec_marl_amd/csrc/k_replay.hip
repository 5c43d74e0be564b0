// f3: device-resident replay ring buffer + policy-output marshalling.
//   * k_replay_store: the E transitions of one vectorised step appended to the ring in env order
//     (= E consecutive buffer.py store_transition calls, BUF:16-25), all seven arrays in ONE launch,
//     straight from the step kernel's outputs (obs / reward / metrics[:,0] / the NOMA mask);
//   * k_replay_sample: sample_buffer (BUF:27-37) -- gather of a batch of rows, indices injected
//     (parity) or Philox (production);
//   * k_marshal_actions: policy outputs -> env action, pairing power and the replay's action row
//     (TRAIN:1386-1396, 1601-1608, 1776-1784).
// Pure HBM copies: segment-major thread mapping so reads and writes of every array are coalesced.
#include "risvec_launch.hpp"

namespace risvec {
namespace {

constexpr uint32_t kSiteReplay = 8;

struct StoreArgs {
    RisVecReplay rb;
    long long cursor;            // mem_cntr before this call
    int n;
    const float* state; const float* action; const float* power_raw; const float* probs;   // action == nullptr: row built from the policy outputs
    const float* reward_g; int rg_stride; const float* reward_l;
    const float* state_; const uint8_t* done; int done_all; const uint8_t* mask;
    float* carry;                // optional [n, input_shape*n_agents]: receives a copy of state_
};

// Segment-major copy: word g of a source array [n, len] lands at row (cursor + g / len) % mem_size.
// The n rows are consecutive in the ring, so there is at most one wrap: words before it go to
// head + g, the rest to g - (words before the wrap) -- no division on the copy path.  With every row
// length a multiple of 4 (n_agents % 4 == 0) a lane moves 16 bytes; the two scalar-per-transition
// arrays (reward_g, done) are handled by the lanes after the vector part.
template <int VEC> struct Pack { using F = float; using U = uint8_t; };
template <> struct Pack<4> { using F = float4; using U = uchar4; };

template <int VEC>
__device__ __forceinline__ typename Pack<VEC>::F widen(typename Pack<VEC>::U m);
template <> __device__ __forceinline__ float widen<1>(uint8_t m) { return m ? 1.0f : 0.0f; }
template <> __device__ __forceinline__ float4 widen<4>(uchar4 m) {
    return make_float4(m.x ? 1.0f : 0.0f, m.y ? 1.0f : 0.0f, m.z ? 1.0f : 0.0f, m.w ? 1.0f : 0.0f);
}
template <int VEC> __device__ __forceinline__ typename Pack<VEC>::F ones();
template <> __device__ __forceinline__ float ones<1>() { return 1.0f; }
template <> __device__ __forceinline__ float4 ones<4>() { return make_float4(1.0f, 1.0f, 1.0f, 1.0f); }

// The ring is write-once data (a learner samples it much later): its stores carry the non-temporal hint so that
// 30 MB per step of transitions do not push the env's h_r / theta working set out of the Infinity Cache.
#ifndef RISVEC_RING_DEFAULT_POLICY
__device__ __forceinline__ void ring_st(float* p, float v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void ring_st(float4* p, float4 v) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(p));
}
#else
__device__ __forceinline__ void ring_st(float* p, float v) { *p = v; }
__device__ __forceinline__ void ring_st(float4* p, float4 v) { *p = v; }
#endif

template <int VEC>
__global__ void __launch_bounds__(kBlock)
k_replay_store(StoreArgs A) {
    RISVEC_ARGS_IN_ONE_TRIP("s"(A.n), "s"(A.cursor), "s"(A.rb.mem_size), "s"(A.rb.n_agents), "s"(A.rb.input_shape),
                            "s"(A.rb.n_actions), "s"(A.state), "s"(A.action), "s"(A.power_raw), "s"(A.probs), "s"(A.reward_l),
                            "s"(A.state_), "s"(A.mask), "s"(A.carry), "s"(A.rb.state_memory), "s"(A.rb.action_memory),
                            "s"(A.rb.reward_local_memory), "s"(A.rb.new_state_memory), "s"(A.rb.mask_memory));
    using F = typename Pack<VEC>::F;
    using U = typename Pack<VEC>::U;
    const RisVecReplay& rb = A.rb;
    const long long S = (long long)rb.input_shape * rb.n_agents, Ac = (long long)rb.n_actions * rb.n_agents;
    const long long L = rb.n_agents, M = (long long)rb.n_agents * rb.n_agents, n = A.n;
    // segment boundaries in the flat word space: state | action | reward_l | state_ | mask || reward_g | done
    const long long b0 = n * S, b1 = b0 + n * Ac, b2 = b1 + n * L, b3 = b2 + n * S, b4 = b3 + n * M;
    const long long units = b4 / VEC;                          // vector part (VEC divides every row length)
    const long long gid = (long long)blockIdx.x * kBlock + threadIdx.x;
    const long long head = A.cursor % rb.mem_size;             // ring row of transition 0 (uniform)
    const long long rows_to_wrap = rb.mem_size - head;         // transitions that fit before the wrap
    auto dest = [&](long long g, long long len) {
        const long long before = rows_to_wrap * len;
        return g < before ? head * len + g : g - before;
    };
    if (gid < units) {
        const long long w = gid * VEC;
        if (w < b0) {
            ring_st(reinterpret_cast<F*>(rb.state_memory + dest(w, S)), *reinterpret_cast<const F*>(A.state + w));
        } else if (w < b1) {
            const long long g = w - b0;
            if (A.action) {
                ring_st(reinterpret_cast<F*>(rb.action_memory + dest(g, Ac)), *reinterpret_cast<const F*>(A.action + g));
            } else {
                // the row risvec_marshal_actions would have built (TRAIN:1386-1390, 1776-1784), straight from the policy
                // outputs: per agent [probs_i with zero diagonal | raw power_i]
                float o[VEC];
                const unsigned W = (unsigned)rb.n_agents + 2, Vn = (unsigned)rb.n_agents;
                // one 32-bit division per lane (n * Ac < 2^32 is checked by the launcher), then walk the VEC words
                unsigned ev = (unsigned)g / W, k = (unsigned)g - ev * W, v = ev % Vn;
#pragma unroll
                // the ADDRESS is selected, not the load: VEC independent requests and one wait (a load per branch of the
                // select made VEC dependent memory round trips for a third of the kernel's lanes)
                for (int c = 0; c < VEC; ++c) {
                    const float* src = k < Vn ? A.probs + ((size_t)ev * Vn + k) : A.power_raw + ((size_t)ev * 2 + (k - Vn));
                    const float x = *src;
                    o[c] = k == v ? 0.0f : x;                  // k == v only happens inside the probs part (v < Vn)
                    if (++k == W) { k = 0; ++ev; v = v + 1 == Vn ? 0 : v + 1; }
                }
                ring_st(reinterpret_cast<F*>(rb.action_memory + dest(g, Ac)), *reinterpret_cast<const F*>(o));
            }
        } else if (w < b2) {
            const long long g = w - b1;
            ring_st(reinterpret_cast<F*>(rb.reward_local_memory + dest(g, L)), *reinterpret_cast<const F*>(A.reward_l + g));
        } else if (w < b3) {
            const long long g = w - b2;
            const F v = *reinterpret_cast<const F*>(A.state_ + g);
            ring_st(reinterpret_cast<F*>(rb.new_state_memory + dest(g, S)), v);
            if (A.carry) *reinterpret_cast<F*>(A.carry + g) = v;      // next step's `state`, for free
        } else {
            const long long g = w - b3;
            ring_st(reinterpret_cast<F*>(rb.mask_memory + dest(g, M)),
                    A.mask ? widen<VEC>(*reinterpret_cast<const U*>(A.mask + g)) : ones<VEC>());    // TRAIN:1786-1789
        }
        return;
    }
    const long long t = gid - units;
    if (t < n) {
        rb.reward_global_memory[dest(t, 1)] = A.reward_g[t * A.rg_stride];
    } else if (t < 2 * n) {
        const long long e = t - n;
        rb.terminal_memory[dest(e, 1)] = A.done ? (A.done[e] ? 1 : 0) : (A.done_all ? 1 : 0);
    }
}

struct SampleArgs {
    RisVecReplay rb;
    long long max_mem;
    int batch;
    const int64_t* idx;
    uint64_t seed;
    uint32_t counter;
    float* states; float* actions; float* rewards_g; float* rewards_l; float* states_; uint8_t* dones; float* masks;
    int64_t* idx_out;
};

__device__ __forceinline__ long long sample_row(const SampleArgs& A, long long b) {
    if (A.idx) return A.idx[b];
    const uint32_t x = philox4x32_10((uint32_t)b, 0u, A.counter, kSiteReplay, A.seed).x;
    return (long long)(((unsigned long long)x * (unsigned long long)A.max_mem) >> 32);
}

__global__ void __launch_bounds__(kBlock)
k_replay_sample(SampleArgs A) {
    const RisVecReplay& rb = A.rb;
    const long long S = (long long)rb.input_shape * rb.n_agents, Ac = (long long)rb.n_actions * rb.n_agents;
    const long long L = rb.n_agents, M = (long long)rb.n_agents * rb.n_agents, n = A.batch;
    const long long b0 = n * S, b1 = b0 + n * Ac, b2 = b1 + n * L, b3 = b2 + n * S, b4 = b3 + n * M, b5 = b4 + n,
                    b6 = b5 + n;
    const long long gid = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= b6) return;
    if (gid < b0) {
        const long long b = gid / S, k = gid % S;
        A.states[gid] = rb.state_memory[sample_row(A, b) * S + k];
    } else if (gid < b1) {
        const long long g = gid - b0, b = g / Ac, k = g % Ac;
        A.actions[g] = rb.action_memory[sample_row(A, b) * Ac + k];
    } else if (gid < b2) {
        const long long g = gid - b1, b = g / L, k = g % L;
        A.rewards_l[g] = rb.reward_local_memory[sample_row(A, b) * L + k];
    } else if (gid < b3) {
        const long long g = gid - b2, b = g / S, k = g % S;
        A.states_[g] = rb.new_state_memory[sample_row(A, b) * S + k];
    } else if (gid < b4) {
        const long long g = gid - b3, b = g / M, k = g % M;
        A.masks[g] = rb.mask_memory[sample_row(A, b) * M + k];
    } else if (gid < b5) {
        const long long b = gid - b4;
        const long long r = sample_row(A, b);
        A.rewards_g[b] = rb.reward_global_memory[r];
        if (A.idx_out) A.idx_out[b] = r;
    } else {
        const long long b = gid - b5;
        A.dones[b] = rb.terminal_memory[sample_row(A, b)];
    }
}

// one lane per word of the replay action row [E, V, V+2]; the lane of word V of each agent also
// writes that agent's env action and pairing power
__global__ void __launch_bounds__(kBlock)
k_marshal_actions(int E, int V, const float* power_raw, const float* probs, float floor_eff, float* action_env,
                  float* p_off01, float* action_store) {
    const unsigned gid = blockIdx.x * kBlock + threadIdx.x;   // E*V*(V+2) < 2^31 is checked by the API
    const unsigned W = V + 2;
    if (gid >= (unsigned)E * V * W) return;
    const unsigned k = gid % W, ev = gid / W, v = ev % V, e = ev / V;
    if (action_store) {
        float out;
        if (k < (unsigned)V) out = k == v ? 0.0f : probs[ev * V + k];              // np.fill_diagonal(., 0), TRAIN:1390
        else out = power_raw[ev * 2 + (k - V)];                          // raw policy output, TRAIN:1777-1782
        action_store[gid] = out;
    }
    if (k == (unsigned)V) {
        const float a0 = power_raw[ev * 2], a1 = power_raw[ev * 2 + 1];
        const float m0 = (fminf(fmaxf(a0, -0.999f), 0.999f) + 1.0f) / 2.0f;    // TRAIN:1603-1605
        const float m1 = (fminf(fmaxf(a1, -0.999f), 0.999f) + 1.0f) / 2.0f;
        if (action_env) {
            action_env[(e * 2 + 0) * V + v] = m0;
            action_env[(e * 2 + 1) * V + v] = fmaxf(m1, floor_eff);             // TRAIN:1606-1608
        }
        if (p_off01) p_off01[ev] = m0;                                         // TRAIN:1393-1396
    }
}

// Even V: one lane per float2 of the replay action row [E, V, V+2] (8-byte accesses, fully coalesced
// on the write side; the probs / power reads are contiguous per agent); the lane holding an agent's
// power pair also writes that agent's env action and pairing power.
__global__ void __launch_bounds__(kBlock)
k_marshal_pairs(int E, int V, const float* power_raw, const float* probs, float floor_eff, float* action_env,
                float* p_off01, float* action_store) {
    RISVEC_ARGS_IN_ONE_TRIP("s"(E), "s"(V), "s"(power_raw), "s"(probs), "s"(action_env), "s"(p_off01), "s"(action_store));
    const unsigned gid = blockIdx.x * kBlock + threadIdx.x;    // E*V*(V+2)/2 < 2^31 is checked by the API
    const unsigned H = (unsigned)V / 2 + 1;                    // float2 per row
    if (gid >= (unsigned)E * V * H) return;
    const unsigned k2 = gid % H, ev = gid / H;
    if (k2 < (unsigned)V / 2) {
        if (!action_store) return;
        const unsigned v = ev % V;
        float2 p = *reinterpret_cast<const float2*>(probs + (size_t)ev * V + 2 * k2);
        if (2 * k2 == v) p.x = 0.0f;                                   // np.fill_diagonal(., 0), TRAIN:1390
        if (2 * k2 + 1 == v) p.y = 0.0f;
        *reinterpret_cast<float2*>(action_store + (size_t)gid * 2) = p;
        return;
    }
    const unsigned v = ev % V, e = ev / V;
    const float2 pw = *reinterpret_cast<const float2*>(power_raw + (size_t)ev * 2);
    if (action_store) *reinterpret_cast<float2*>(action_store + (size_t)gid * 2) = pw;   // raw output, TRAIN:1777-1782
    const float m0 = (fminf(fmaxf(pw.x, -0.999f), 0.999f) + 1.0f) / 2.0f;       // TRAIN:1603-1605
    const float m1 = (fminf(fmaxf(pw.y, -0.999f), 0.999f) + 1.0f) / 2.0f;
    if (action_env) {
        action_env[((size_t)e * 2 + 0) * V + v] = m0;
        action_env[((size_t)e * 2 + 1) * V + v] = fmaxf(m1, floor_eff);          // TRAIN:1606-1608
    }
    if (p_off01) p_off01[ev] = m0;                                              // TRAIN:1393-1396
}

}  // namespace

hipError_t launch_replay_store(const RisVecReplay& rb, long long cursor, int n, const float* state, const float* action,
                               const float* power_raw, const float* probs, const float* reward_g, int rg_stride, const float* reward_l, const float* state_,
                               const uint8_t* done, int done_all, const uint8_t* mask, float* carry, hipStream_t st) {
    const long long S = (long long)rb.input_shape * rb.n_agents, Ac = (long long)rb.n_actions * rb.n_agents;
    const long long L = rb.n_agents, M = L * L;
    const long long words = (long long)n * (2 * S + Ac + L + M);
    if (!action && (long long)n * Ac >= (1LL << 32)) return hipErrorInvalidValue;   // 32-bit word index in the policy-output form
    StoreArgs a{rb, cursor, n, state, action, power_raw, probs, reward_g, rg_stride, reward_l, state_, done, done_all, mask, carry};
    const bool vec = S % 4 == 0 && Ac % 4 == 0 && L % 4 == 0;      // then M = L*L is too
    const long long threads = (vec ? words / 4 : words) + 2LL * n;
    const dim3 grid((unsigned)((threads + kBlock - 1) / kBlock));
    if (vec) hipLaunchKernelGGL(k_replay_store<4>, grid, dim3(kBlock), 0, st, a);
    else hipLaunchKernelGGL(k_replay_store<1>, grid, dim3(kBlock), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_replay_sample(const RisVecReplay& rb, long long max_mem, int batch, const int64_t* idx, uint64_t seed,
                                uint32_t counter, float* states, float* actions, float* rewards_g, float* rewards_l,
                                float* states_, uint8_t* dones, float* masks, int64_t* idx_out, hipStream_t st) {
    const long long S = (long long)rb.input_shape * rb.n_agents, Ac = (long long)rb.n_actions * rb.n_agents;
    const long long words = (long long)batch * (2 * S + Ac + rb.n_agents + (long long)rb.n_agents * rb.n_agents + 2);
    SampleArgs a{rb, max_mem, batch, idx, seed, counter, states, actions, rewards_g, rewards_l, states_, dones, masks,
                 idx_out};
    hipLaunchKernelGGL(k_replay_sample, dim3((unsigned)((words + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_marshal_actions(int E, int V, const float* power_raw, const float* probs, float floor_eff,
                                  float* action_env, float* p_off01, float* action_store, hipStream_t st) {
    if (V % 2 == 0) {
        const long long pairs = (long long)E * V * (V / 2 + 1);
        hipLaunchKernelGGL(k_marshal_pairs, dim3((unsigned)((pairs + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, E, V,
                           power_raw, probs, floor_eff, action_env, p_off01, action_store);
        return hipGetLastError();
    }
    const long long words = (long long)E * V * (V + 2);
    hipLaunchKernelGGL(k_marshal_actions, dim3((unsigned)((words + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, E, V,
                       power_raw, probs, floor_eff, action_env, p_off01, action_store);
    return hipGetLastError();
}

}  // namespace risvec
