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
    const float* state; const float* action; const float* reward_g; int rg_stride; const float* reward_l;
    const float* state_; const uint8_t* done; int done_all; const uint8_t* mask;
};

__global__ void __launch_bounds__(kBlock)
k_replay_store(StoreArgs A) {
    const RisVecReplay& rb = A.rb;
    const long long S = (long long)rb.input_shape * rb.n_agents, Ac = (long long)rb.n_actions * rb.n_agents;
    const long long L = rb.n_agents, M = (long long)rb.n_agents * rb.n_agents, n = A.n;
    // segment boundaries in the flat word space: state | action | reward_l | state_ | mask | reward_g | done
    const long long b0 = n * S, b1 = b0 + n * Ac, b2 = b1 + n * L, b3 = b2 + n * S, b4 = b3 + n * M, b5 = b4 + n,
                    b6 = b5 + n;
    const long long gid = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= b6) return;
    auto row_of = [&](long long e) { return (A.cursor + e) % rb.mem_size; };
    if (gid < b0) {
        const long long e = gid / S, k = gid % S;
        rb.state_memory[row_of(e) * S + k] = A.state[gid];
    } else if (gid < b1) {
        const long long g = gid - b0, e = g / Ac, k = g % Ac;
        rb.action_memory[row_of(e) * Ac + k] = A.action[g];
    } else if (gid < b2) {
        const long long g = gid - b1, e = g / L, k = g % L;
        rb.reward_local_memory[row_of(e) * L + k] = A.reward_l[g];
    } else if (gid < b3) {
        const long long g = gid - b2, e = g / S, k = g % S;
        rb.new_state_memory[row_of(e) * S + k] = A.state_[g];
    } else if (gid < b4) {
        const long long g = gid - b3, e = g / M, k = g % M;
        rb.mask_memory[row_of(e) * M + k] = A.mask ? (A.mask[g] ? 1.0f : 0.0f) : 1.0f;    // TRAIN:1786-1789
    } else if (gid < b5) {
        const long long e = gid - b4;
        rb.reward_global_memory[row_of(e)] = A.reward_g[e * A.rg_stride];
    } else {
        const long long e = gid - b5;
        rb.terminal_memory[row_of(e)] = A.done ? (A.done[e] ? 1 : 0) : (A.done_all ? 1 : 0);
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
    const long long gid = (long long)blockIdx.x * kBlock + threadIdx.x;
    const int W = V + 2;
    if (gid >= (long long)E * V * W) return;
    const int k = (int)(gid % W);
    const long long ev = gid / W;
    const int v = (int)(ev % V);
    const long long e = ev / V;
    if (action_store) {
        float out;
        if (k < V) out = k == v ? 0.0f : probs[ev * V + k];              // np.fill_diagonal(., 0), TRAIN:1390
        else out = power_raw[ev * 2 + (k - V)];                          // raw policy output, TRAIN:1777-1782
        action_store[gid] = out;
    }
    if (k == V) {
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

}  // namespace

hipError_t launch_replay_store(const RisVecReplay& rb, long long cursor, int n, const float* state, const float* action,
                               const float* reward_g, int rg_stride, const float* reward_l, const float* state_,
                               const uint8_t* done, int done_all, const uint8_t* mask, hipStream_t st) {
    const long long S = (long long)rb.input_shape * rb.n_agents, Ac = (long long)rb.n_actions * rb.n_agents;
    const long long words = (long long)n * (2 * S + Ac + rb.n_agents + (long long)rb.n_agents * rb.n_agents + 2);
    StoreArgs a{rb, cursor, n, state, action, reward_g, rg_stride, reward_l, state_, done, done_all, mask};
    hipLaunchKernelGGL(k_replay_store, dim3((unsigned)((words + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, a);
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
    const long long words = (long long)E * V * (V + 2);
    hipLaunchKernelGGL(k_marshal_actions, dim3((unsigned)((words + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, E, V,
                       power_raw, probs, floor_eff, action_env, p_off01, action_store);
    return hipGetLastError();
}

}  // namespace risvec
