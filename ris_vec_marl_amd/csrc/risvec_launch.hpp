// Internal launcher declarations (one per kernel family); argument validation lives
// in risvec_api.hip, these only compute the grid and launch.
#pragma once

#include <hip/hip_runtime.h>

#include "risvec.h"
#include "risvec_dev.hpp"

namespace risvec {

hipError_t launch_reset(const RisVecState& s, const RisVecParams& p, const int32_t* spawn_ints,
                        const int32_t* buf0, uint64_t seed, uint32_t counter, hipStream_t st);
hipError_t launch_mobility(const RisVecState& s, const RisVecParams& p, const float* u_turn,
                           int32_t* n_used, uint64_t seed, uint32_t counter, hipStream_t st);
hipError_t launch_geometry(const RisVecState& s, const RisVecParams& p, hipStream_t st);
hipError_t launch_gain_3gpp(const RisVecState& s, const RisVecParams& p, int32_t model,
                            const float* u_los, const float* z_shadow, const float* small,
                            uint64_t seed, uint32_t counter, hipStream_t st);
hipError_t launch_set_phase(const RisVecState& s, const float* angle, hipStream_t st);
hipError_t launch_random_phase(const RisVecState& s, const int32_t* idx, uint64_t seed,
                               uint32_t counter, hipStream_t st);

hipError_t launch_gain(const RisVecState& s, const RisVecParams& p, hipStream_t st);
struct StepRing;
hipError_t launch_step(const RisVecState& s, const RisVecParams& p, const float* action,
                       const int32_t* partner, const int32_t* n_groups, const int32_t* arrivals,
                       uint64_t seed, uint32_t counter, uint32_t flags, bool fused, hipStream_t st,
                       const StepRing* ring = nullptr);

hipError_t launch_data_rate(const RisVecState& s, const RisVecParams& p, const float* p_off,
                            const int32_t* partner, const int32_t* n_groups, float* rate_out,
                            hipStream_t st);

hipError_t launch_colsum(const RisVecState& s, hipStream_t st);
hipError_t launch_bcd(const RisVecState& s, const RisVecParams& p, int32_t* idx_out, bool reuse_colsum,
                      bool reuse_s, bool reuse_idx, bool write_theta, hipStream_t st);
hipError_t launch_step_fused_bcd(const RisVecState& s, const RisVecParams& p, const float* action,
                                 const int32_t* partner, const int32_t* n_groups,
                                 const int32_t* arrivals, uint64_t seed, uint32_t counter,
                                 uint32_t flags, hipStream_t st);

hipError_t launch_sarl_step(const RisVecState& s, const RisVecSarlParams& p, const float* action_power,
                            const float* action_phase, const int32_t* arrivals, uint64_t seed,
                            uint32_t counter, uint32_t flags, hipStream_t st);

long long noma_scratch_bytes(int n_envs, int n_veh);
hipError_t launch_noma_begin_episode(const RisVecNomaState& ns, hipStream_t st);
hipError_t launch_noma_mask(const RisVecNomaState& ns, const float* gain, const double* gdb15, double q_now,
                            int K_now, hipStream_t st);
hipError_t launch_noma_group(const RisVecNomaState& ns, const RisVecNomaParams& p, const float* gain,
                             const double* gdb12, const float* p01, int p01_raw, int use_mask, int K_back,
                             const double* tau_back, const float* prev_global, int prev_stride, int i_step,
                             const float* u_unstick, uint64_t seed, uint32_t counter, int32_t* info_out,
                             hipStream_t st);
hipError_t launch_noma_flush(const RisVecNomaState& ns, float decay, hipStream_t st);

hipError_t launch_replay_store(const RisVecReplay& rb, long long cursor, int n, const float* state, const float* action,
                               const float* power_raw, const float* probs, const float* reward_g, int rg_stride, const float* reward_l, const float* state_,
                               const uint8_t* done, int done_all, const uint8_t* mask, float* carry, hipStream_t st);
hipError_t launch_replay_sample(const RisVecReplay& rb, long long max_mem, int batch, const int64_t* idx, uint64_t seed,
                                uint32_t counter, float* states, float* actions, float* rewards_g, float* rewards_l,
                                float* states_, uint8_t* dones, float* masks, int64_t* idx_out, hipStream_t st);
hipError_t launch_marshal_actions(int E, int V, const float* power_raw, const float* probs, float floor_eff,
                                  float* action_env, float* p_off01, float* action_store, hipStream_t st);

hipError_t launch_policy_sample(int E, int V, long long env_offset, const float* heads, const uint8_t* mask,
                                const float* tau, const uint8_t* hard, const float* eps, const float* expo, uint64_t seed,
                                uint32_t counter, float floor_eff, float* power_raw, float* probs, float* onehot,
                                float* action_env, float* p_off01, float* action_store, hipStream_t st);

hipError_t launch_policy_layer1(int E, int V, int IN, int F, const float* obs, const float* W1, const float* b1,
                                const float* lw, const float* lb, float* out, hipStream_t st);
hipError_t launch_policy_layer1_split16(int E, int V, int IN, int F, const float* obs, const float* W1, const float* b1,
                                        const float* lw, const float* lb, void* out16, hipStream_t st);
bool policy_mlp_supported(int IN, int F1, int F2, int H);
hipError_t launch_policy_mlp(int E, int V, int IN, int F1, int F2, int H, const float* obs, const float* G, const void* W1F,
                             const void* W2f, const float* gscale, const float* b2, const float* ln2w, const float* ln2b,
                             const void* WhF, const float* hscale, const float* bh, float* heads, hipStream_t st);
hipError_t launch_policy_heads(int E, int V, int F, int H, const float* g, const float* b2, const float* lw,
                               const float* lb, const float* Wh, const float* bh, float* heads, hipStream_t st);

int episode_partial_rows(int E);
hipError_t launch_episode_clear(int E, int V, double* acc, hipStream_t st);
hipError_t launch_episode_accumulate(int E, int V, const float* metrics, const float* reward, const float* power_w,
                                     float user_clip, double* acc, hipStream_t st);
hipError_t launch_episode_summary(int E, int V, int n_steps, const double* acc, const float* metrics, double* per_env,
                                  double* partial, double* summary, hipStream_t st);

// bytes per env row of state.theta_idx: the candidate index of every theta element, padded to a multiple of 32
// (four 8-element tiles: the pair sweep reads and writes the indices of four tiles per request)
__host__ __device__ inline int theta_idx_stride(int n_ris) { return (n_ris + 31) / 32 * 32; }
hipError_t launch_theta_from_index(const RisVecState& s, hipStream_t st);

// risvec_last_kernel(): the launchers of the step path and the BCD sweep name the kernel they dispatched (per thread)
void note_kernel(const char* fmt, ...);

inline Dims dims_of(const RisVecState& s) {
    return Dims{s.n_envs, s.n_veh, s.n_ris, s.control_bit, (long long)s.env_offset};
}

}  // namespace risvec
