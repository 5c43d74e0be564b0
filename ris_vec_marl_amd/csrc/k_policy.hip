// f3, batched choose_action: the sampling epilogue of the reference's SAC policy
// (sac_agent.py, SAC below) for every (env, agent) at once, fused with the marshalling of
// marl_train_bcd.py (TRAIN).  The three small GEMMs of PolicyNetwork.forward (SAC:62-78) are library
// work (rocBLAS through torch.bmm, see policy.py); this kernel takes their output -- per agent a row
// (mu[2], log_std[2], intent_logits[V]) -- and does everything after it in one pass:
//   log_std clamp (SAC:72), Normal(mu, std).sample() and tanh (SAC:83-86), the feasibility mask on the
//   logits incl. the all-zero-row rule (SAC:91-104), the soft Gumbel-softmax (SAC:110-113), the arg-max
//   one-hot of choose_action (SAC:215-216), and -- optionally -- the env action, pairing power and replay
//   action row of TRAIN:1386-1396, 1601-1608, 1776-1784 (what k_marshal_* would do in a second launch).
// Draws are injected (parity) or Philox.  One lane per (env, agent); V <= 64.
#include <cfloat>

#include "risvec_launch.hpp"

namespace risvec {
namespace {

constexpr uint32_t kSitePolicyEps = 9, kSitePolicyGumbel = 10;

struct PolicyArgs {
    int E, V;
    long long env_offset;
    const float* heads;        // [V, E, 4 + V]
    const uint8_t* mask;       // [E, V, V] or NULL
    const float* tau;          // [V]
    const float* eps;          // [E, V, 2] or NULL
    const float* expo;         // [E, V, V] or NULL
    uint64_t seed;
    uint32_t counter;
    float floor_eff;
    float* power_raw;          // [E, V, 2]
    float* probs;              // [E, V, V]
    float* onehot;             // [E, V, V] or NULL
    float* action_env;         // [E, 2, V] or NULL
    float* p_off01;            // [E, V] or NULL
    float* action_store;       // [E, V, V + 2] or NULL
};

__global__ void __launch_bounds__(kBlock)
k_policy_sample(PolicyArgs A) {
    const long long gid = (long long)blockIdx.x * kBlock + threadIdx.x;
    const int V = A.V, H = 4 + V;
    if (gid >= (long long)A.E * V) return;
    const long long e = gid / V;
    const int v = (int)(gid % V);
    const float* h = A.heads + ((long long)v * A.E + e) * H;
    const uint32_t genv = (uint32_t)(A.env_offset + e);
    // ---- continuous head: x = mu + std * eps, power = tanh(x)  (SAC:72, 83-86) ------------------------
    float e0, e1;
    if (A.eps) { e0 = A.eps[gid * 2]; e1 = A.eps[gid * 2 + 1]; }
    else {
        const uint4 r = philox4x32_10(genv, (uint32_t)v, A.counter, kSitePolicyEps, A.seed);
        const float2 n = normal2(r.x, r.y);
        e0 = n.x; e1 = n.y;
    }
    const float ls0 = fminf(fmaxf(h[2], -20.0f), 2.0f), ls1 = fminf(fmaxf(h[3], -20.0f), 2.0f);
    const float p0 = tanhf(e0 * expf(ls0) + h[0]), p1 = tanhf(e1 * expf(ls1) + h[1]);
    A.power_raw[gid * 2] = p0;
    A.power_raw[gid * 2 + 1] = p1;
    // ---- discrete head: masked logits, soft Gumbel-softmax  (SAC:91-113) -------------------------------
    const uint8_t* mrow = A.mask ? A.mask + gid * V : nullptr;
    bool any_open = false;
    if (mrow) for (int k = 0; k < V; ++k) any_open = any_open || mrow[k] != 0;
    const bool use_mask = mrow && any_open;              // an all-zero row is opened up (SAC:97-100)
    const float neg_large = -FLT_MAX / 2.0f;             // torch.finfo(float32).min / 2  (SAC:103)
    const float tau_v = A.tau[v];
    float* y = A.probs + gid * V;
    float zmax = -INFINITY;
    int arg = 0;
    uint4 r = make_uint4(0, 0, 0, 0);
    for (int k = 0; k < V; ++k) {
        float ex;
        if (A.expo) ex = A.expo[gid * V + k];
        else {
            if ((k & 3) == 0) r = philox4x32_10(genv, (uint32_t)v, A.counter, kSitePolicyGumbel + 0x100u * (k >> 2), A.seed);
            const uint32_t x = (k & 3) == 0 ? r.x : (k & 3) == 1 ? r.y : (k & 3) == 2 ? r.z : r.w;
            ex = -logf(((float)(x >> 8) + 1.0f) * 0x1p-24f);          // Exp(1), u in (0, 1]
        }
        const float ml = (use_mask && mrow[k] == 0) ? neg_large : h[4 + k];
        const float z = (ml + -logf(ex)) / tau_v;                     // (logits + gumbel) / tau
        y[k] = z;
        if (z > zmax) { zmax = z; arg = k; }
    }
    float sum = 0.0f;
    for (int k = 0; k < V; ++k) { const float t = expf(y[k] - zmax); y[k] = t; sum += t; }
    for (int k = 0; k < V; ++k) {
        const float pk = y[k] / sum;
        y[k] = pk;
        if (A.onehot) A.onehot[gid * V + k] = k == arg ? 1.0f : 0.0f;   // choose_action, SAC:215-216
        if (A.action_store) A.action_store[gid * (V + 2) + k] = k == v ? 0.0f : pk;   // TRAIN:1390, 1776-1784
    }
    // ---- marshalling (optional)  TRAIN:1391-1396, 1601-1608 ---------------------------------------------
    if (A.action_store) {
        A.action_store[gid * (V + 2) + V] = p0;
        A.action_store[gid * (V + 2) + V + 1] = p1;
    }
    const float m0 = (fminf(fmaxf(p0, -0.999f), 0.999f) + 1.0f) / 2.0f;
    const float m1 = (fminf(fmaxf(p1, -0.999f), 0.999f) + 1.0f) / 2.0f;
    if (A.action_env) {
        A.action_env[(e * 2 + 0) * V + v] = m0;
        A.action_env[(e * 2 + 1) * V + v] = fmaxf(m1, A.floor_eff);
    }
    if (A.p_off01) A.p_off01[gid] = m0;
}

}  // namespace

hipError_t launch_policy_sample(int E, int V, long long env_offset, const float* heads, const uint8_t* mask,
                                const float* tau, const float* eps, const float* expo, uint64_t seed, uint32_t counter,
                                float floor_eff, float* power_raw, float* probs, float* onehot, float* action_env,
                                float* p_off01, float* action_store, hipStream_t st) {
    PolicyArgs a{E, V, env_offset, heads, mask, tau, eps, expo, seed, counter, floor_eff, power_raw, probs, onehot,
                 action_env, p_off01, action_store};
    const long long n = (long long)E * V;
    hipLaunchKernelGGL(k_policy_sample, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, a);
    return hipGetLastError();
}

}  // namespace risvec
