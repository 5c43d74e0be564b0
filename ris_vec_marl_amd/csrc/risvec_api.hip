// C ABI of librisvec.so (declared in include/risvec.h): argument validation, error
// strings, and dispatch to the kernel launchers.  Nothing here computes on the CPU:
// every entry point either launches HIP kernels or fails with an error code.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "risvec_launch.hpp"
#include "risvec_step.hpp"

namespace {

thread_local char g_err[512] = "";
thread_local char g_kernel[160] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

#define REQ_PTR(ptr, name)                                                                   \
    do {                                                                                     \
        if ((ptr) == nullptr) return fail(RISVEC_ERR_ARG, "%s: %s is NULL", fn, name);       \
        if (!aligned16(ptr)) return fail(RISVEC_ERR_ARG, "%s: %s is not 16-byte aligned", fn, name); \
    } while (0)
#define OPT_PTR(ptr, name)                                                                   \
    do {                                                                                     \
        if ((ptr) != nullptr && !aligned16(ptr))                                             \
            return fail(RISVEC_ERR_ARG, "%s: %s is not 16-byte aligned", fn, name);          \
    } while (0)

int check_common(const char* fn, const RisVecState* s, const RisVecParams* p) {
    if (!s) return fail(RISVEC_ERR_ARG, "%s: state is NULL", fn);
    if (s->abi_version != RISVEC_ABI_VERSION || s->struct_bytes != sizeof(RisVecState))
        return fail(RISVEC_ERR_ARG, "%s: RisVecState ABI mismatch (version %u/%u, bytes %u/%zu)", fn,
                    s->abi_version, (unsigned)RISVEC_ABI_VERSION, s->struct_bytes, sizeof(RisVecState));
    if (p && (p->abi_version != RISVEC_ABI_VERSION || p->struct_bytes != sizeof(RisVecParams)))
        return fail(RISVEC_ERR_ARG, "%s: RisVecParams ABI mismatch (version %u/%u, bytes %u/%zu)", fn,
                    p->abi_version, (unsigned)RISVEC_ABI_VERSION, p->struct_bytes, sizeof(RisVecParams));
    if (s->n_envs < 1) return fail(RISVEC_ERR_SHAPE, "%s: n_envs=%d must be >= 1", fn, s->n_envs);
    if (s->n_veh < 1 || s->n_veh > RISVEC_MAX_VEH)
        return fail(RISVEC_ERR_SHAPE, "%s: n_veh=%d outside [1,%d]", fn, s->n_veh, RISVEC_MAX_VEH);
    if (s->n_ris < 1 || s->n_ris > 2048)
        return fail(RISVEC_ERR_SHAPE, "%s: n_ris=%d outside [1,2048]", fn, s->n_ris);
    if (s->control_bit < 0 || s->control_bit > 6)
        return fail(RISVEC_ERR_SHAPE, "%s: control_bit=%d outside [0,6]", fn, s->control_bit);
    if ((long long)s->n_envs * s->n_veh * s->n_ris > (1LL << 40))
        return fail(RISVEC_ERR_SHAPE, "%s: E*V*M too large", fn);
    if (s->env_offset < 0 || s->env_offset + s->n_envs > 0xFFFFFFFFLL)
        return fail(RISVEC_ERR_SHAPE, "%s: env_offset+n_envs must fit 32 bits", fn);
    if (p && (p->n_lanes < 1 || p->n_lanes > RISVEC_MAX_LANES))
        return fail(RISVEC_ERR_SHAPE, "%s: n_lanes=%d outside [1,%d]", fn, p->n_lanes, RISVEC_MAX_LANES);
    return RISVEC_OK;
}

int finish(const char* fn, hipError_t err) {
    if (err != hipSuccess) return fail(RISVEC_ERR_LAUNCH, "%s: %s", fn, hipGetErrorString(err));
    return RISVEC_OK;
}

int check_step(const char* fn, const RisVecState* s, const float* action, const int32_t* partner,
               const int32_t* n_groups, const int32_t* arrivals, uint32_t flags, bool fused) {
    REQ_PTR(action, "action"); REQ_PTR(partner, "partner"); REQ_PTR(n_groups, "n_groups");
    OPT_PTR(arrivals, "arrivals");
    REQ_PTR(s->gain, "state.gain"); REQ_PTR(s->data_buf, "state.data_buf"); REQ_PTR(s->mec_q, "state.mec_q");
    REQ_PTR(s->rate, "state.rate"); REQ_PTR(s->data_t, "state.data_t"); REQ_PTR(s->data_p, "state.data_p");
    REQ_PTR(s->reward, "state.reward"); REQ_PTR(s->over_power, "state.over_power");
    REQ_PTR(s->metrics, "state.metrics");
    if (flags & RISVEC_STEP_OBS) REQ_PTR(s->obs, "state.obs");
    if (flags & RISVEC_STEP_POWER_W) REQ_PTR(s->power_w, "state.power_w");
    if (flags & ~(uint32_t)(RISVEC_STEP_METRICS | RISVEC_STEP_POWER_W | RISVEC_STEP_POLICY_ACTION | RISVEC_STEP_OBS |
                            RISVEC_STEP_REUSE_COLSUM | RISVEC_STEP_REUSE_SSUM | RISVEC_STEP_REUSE_IDX | RISVEC_STEP_STEER |
                            RISVEC_STEP_THETA_BY_INDEX))
        return fail(RISVEC_ERR_ARG, "%s: unknown flag bits 0x%x", fn, flags);
    if (flags & RISVEC_STEP_STEER) {
        if (!fused) return fail(RISVEC_ERR_ARG, "%s: RISVEC_STEP_STEER needs the fused entry points", fn);
        REQ_PTR(s->z_r, "state.z_r");
    }
    if (fused) {
        REQ_PTR(s->h_r, "state.h_r"); REQ_PTR(s->theta, "state.theta"); REQ_PTR(s->b, "state.b");
        REQ_PTR(s->pl, "state.pl"); OPT_PTR(s->h_d, "state.h_d");
    }
    return RISVEC_OK;
}

}  // namespace

namespace risvec {
void note_kernel(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
    va_end(ap);
}
}  // namespace risvec

extern "C" {

uint32_t risvec_abi_version(void) { return RISVEC_ABI_VERSION; }

const char* risvec_last_kernel(void) { return g_kernel; }

const char* risvec_last_error(void) { return g_err; }

void risvec_default_params(RisVecParams* p) {
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->abi_version = RISVEC_ABI_VERSION;
    p->struct_bytes = sizeof(RisVecParams);
    p->bandwidth_mhz = 1.0f;                                           // ENV:72
    p->noise_power = (float)(std::pow(10.0, (-174.0 - 30.0) / 10.0) * 1.0e6);   // ENV:74-76
    p->p_max = 1.0f;                                                   // ENV:125
    p->power_scale = 0.7f;                                             // ENV:555
    p->qos_enable = 1; p->r_min_bpshz = 0.20f; p->d_max_s = 0.10f; p->qos_penalty = 5.0f;   // ENV:79-82
    p->time_fast = 1e-3f;                                              // ENV:102
    p->k_cpu = 1e-28f;                                                 // ENV:104
    p->f_local_max = 1.0e9f; p->f_edge_max = 2.0e9f;                   // ENV:108-109
    p->cycles_per_bit = 500.0f; p->cpu_share_floor = 0.10f;            // ENV:111-113
    p->w_d = 0.5f; p->w_e = 3.0f; p->reward_clip = 50.0f;              // ENV:138-143
    p->arrival_rate = 3.0f;                                            // ENV:156
    {   // Poisson(3) CDF, float64 -> float32
        double pk = std::exp(-3.0), cdf = 0.0;
        for (int k = 0; k < RISVEC_POISSON_TABLE; ++k) {
            cdf += pk;
            p->poisson_cdf[k] = (float)(cdf < 1.0 ? cdf : 1.0);
            pk *= 3.0 / (double)(k + 1);
        }
    }
    p->fc_ghz = 3.5f; p->shadow_std_los = 4.0f; p->shadow_std_nlos = 7.0f;   // ENV:186-188
    p->rician_k_db = 0.0f; p->veh_ant_gain = 3.0f;                     // ENV:189, 96
    p->n_lanes = 4;
    p->time_slow = 0.1; p->width = 400.0; p->height = 400.0;           // ENV:101
    const double up[4] = {(400 + 3.5 / 2) / 2.0, (400 + 3.5 + 3.5 / 2) / 2.0, (800 + 3.5 / 2) / 2.0,
                          (800 + 3.5 + 3.5 / 2) / 2.0};                // marl_train_bcd.py:446
    const double dn[4] = {(400 - 3.5 - 3.5 / 2) / 2.0, (400 - 3.5 / 2) / 2.0, (800 - 3.5 - 3.5 / 2) / 2.0,
                          (800 - 3.5 / 2) / 2.0};                      // marl_train_bcd.py:447
    for (int i = 0; i < 4; ++i) {
        p->lanes_up[i] = up[i]; p->lanes_left[i] = up[i];
        p->lanes_down[i] = dn[i]; p->lanes_right[i] = dn[i];
    }
}

int risvec_reset(const RisVecState* s, const RisVecParams* p, const int32_t* spawn_ints,
                 const int32_t* buf0, uint64_t seed, uint32_t counter, risvec_stream_t stream) {
    const char* fn = "risvec_reset";
    if (!p) return fail(RISVEC_ERR_ARG, "%s: params is NULL", fn);
    if (int rc = check_common(fn, s, p)) return rc;
    REQ_PTR(s->pos, "state.pos"); REQ_PTR(s->dir, "state.dir"); REQ_PTR(s->vel, "state.vel");
    REQ_PTR(s->data_buf, "state.data_buf");
    OPT_PTR(spawn_ints, "spawn_ints"); OPT_PTR(buf0, "buf0");
    if ((spawn_ints == nullptr) != (buf0 == nullptr))
        return fail(RISVEC_ERR_ARG, "%s: spawn_ints and buf0 must both be given or both be NULL", fn);
    if (p->n_lanes != 4)
        return fail(RISVEC_ERR_UNSUPPORTED, "%s: the reference spawn rule is defined for 4 lanes per direction", fn);
    return finish(fn, risvec::launch_reset(*s, *p, spawn_ints, buf0, seed, counter, (hipStream_t)stream));
}

int risvec_mobility(const RisVecState* s, const RisVecParams* p, const float* u_turn, int32_t* n_used,
                    uint64_t seed, uint32_t counter, risvec_stream_t stream) {
    const char* fn = "risvec_mobility";
    if (!p) return fail(RISVEC_ERR_ARG, "%s: params is NULL", fn);
    if (int rc = check_common(fn, s, p)) return rc;
    REQ_PTR(s->pos, "state.pos"); REQ_PTR(s->dir, "state.dir"); REQ_PTR(s->vel, "state.vel");
    OPT_PTR(u_turn, "u_turn"); OPT_PTR(n_used, "n_used");
    return finish(fn, risvec::launch_mobility(*s, *p, u_turn, n_used, seed, counter, (hipStream_t)stream));
}

int risvec_geometry(const RisVecState* s, const RisVecParams* p, risvec_stream_t stream) {
    const char* fn = "risvec_geometry";
    if (!p) return fail(RISVEC_ERR_ARG, "%s: params is NULL", fn);
    if (int rc = check_common(fn, s, p)) return rc;
    REQ_PTR(s->pos, "state.pos"); REQ_PTR(s->dist_r, "state.dist_r"); REQ_PTR(s->ang_r, "state.ang_r");
    OPT_PTR(s->z_r, "state.z_r");
    REQ_PTR(s->pl, "state.pl"); REQ_PTR(s->h_r, "state.h_r");
    OPT_PTR(s->c_col, "state.c_col");
    if (int rc = finish(fn, risvec::launch_geometry(*s, *p, (hipStream_t)stream))) return rc;
    if (s->c_col) {                      // keep the BCD column-sum cache in step with h_r
        REQ_PTR(s->b, "state.b");
        return finish(fn, risvec::launch_colsum(*s, (hipStream_t)stream));
    }
    return RISVEC_OK;
}

int risvec_gain(const RisVecState* s, const RisVecParams* p, risvec_stream_t stream) {
    const char* fn = "risvec_gain";
    if (!p) return fail(RISVEC_ERR_ARG, "%s: params is NULL", fn);
    if (int rc = check_common(fn, s, p)) return rc;
    REQ_PTR(s->h_r, "state.h_r"); REQ_PTR(s->theta, "state.theta"); REQ_PTR(s->b, "state.b");
    REQ_PTR(s->pl, "state.pl"); REQ_PTR(s->gain, "state.gain"); OPT_PTR(s->h_d, "state.h_d");
    return finish(fn, risvec::launch_gain(*s, *p, (hipStream_t)stream));
}

int risvec_gain_3gpp(const RisVecState* s, const RisVecParams* p, int32_t model, const float* u_los,
                     const float* z_shadow, const float* small, uint64_t seed, uint32_t counter,
                     risvec_stream_t stream) {
    const char* fn = "risvec_gain_3gpp";
    if (!p) return fail(RISVEC_ERR_ARG, "%s: params is NULL", fn);
    if (int rc = check_common(fn, s, p)) return rc;
    if (model != RISVEC_CH_3GPP_UMI && model != RISVEC_CH_3GPP_UMA && model != RISVEC_CH_OTHER)
        return fail(RISVEC_ERR_ARG, "%s: model=%d is not a 3GPP/other model (use risvec_gain for 'free')", fn, model);
    REQ_PTR(s->pos, "state.pos"); REQ_PTR(s->gain, "state.gain");
    OPT_PTR(u_los, "u_los"); OPT_PTR(z_shadow, "z_shadow"); OPT_PTR(small, "small");
    const int n_inj = (u_los != nullptr) + (z_shadow != nullptr) + (small != nullptr);
    if (n_inj != 0 && n_inj != 3)
        return fail(RISVEC_ERR_ARG, "%s: u_los, z_shadow, small must all be given or all be NULL", fn);
    return finish(fn, risvec::launch_gain_3gpp(*s, *p, model, u_los, z_shadow, small, seed, counter,
                                               (hipStream_t)stream));
}

int risvec_colsum(const RisVecState* s, risvec_stream_t stream) {
    const char* fn = "risvec_colsum";
    if (int rc = check_common(fn, s, nullptr)) return rc;
    REQ_PTR(s->h_r, "state.h_r"); REQ_PTR(s->b, "state.b"); REQ_PTR(s->c_col, "state.c_col");
    return finish(fn, risvec::launch_colsum(*s, (hipStream_t)stream));
}

int risvec_bcd(const RisVecState* s, const RisVecParams* p, int32_t* idx_out, uint32_t flags,
               risvec_stream_t stream) {
    const char* fn = "risvec_bcd";
    if (!p) return fail(RISVEC_ERR_ARG, "%s: params is NULL", fn);
    if (int rc = check_common(fn, s, p)) return rc;
    REQ_PTR(s->h_r, "state.h_r"); REQ_PTR(s->theta, "state.theta"); REQ_PTR(s->b, "state.b");
    REQ_PTR(s->c_col, "state.c_col");
    OPT_PTR(idx_out, "idx_out");
    OPT_PTR(s->s_sum, "state.s_sum");
    OPT_PTR(s->theta_idx, "state.theta_idx");
    if (flags & ~(uint32_t)(RISVEC_BCD_REUSE_COLSUM | RISVEC_BCD_REUSE_SSUM | RISVEC_BCD_REUSE_IDX | RISVEC_BCD_NO_THETA))
        return fail(RISVEC_ERR_ARG, "%s: unknown flag bits 0x%x", fn, flags);
    if ((flags & RISVEC_BCD_NO_THETA) && (!(flags & RISVEC_BCD_REUSE_IDX) || s->control_bit != 3))
        return fail(RISVEC_ERR_ARG, "%s: RISVEC_BCD_NO_THETA needs RISVEC_BCD_REUSE_IDX and control_bit = 3", fn);
    if ((flags & RISVEC_BCD_REUSE_SSUM) && !s->s_sum)
        return fail(RISVEC_ERR_ARG, "%s: RISVEC_BCD_REUSE_SSUM needs state.s_sum", fn);
    if ((flags & RISVEC_BCD_REUSE_IDX) && !s->theta_idx)
        return fail(RISVEC_ERR_ARG, "%s: RISVEC_BCD_REUSE_IDX needs state.theta_idx", fn);
    return finish(fn, risvec::launch_bcd(*s, *p, idx_out, (flags & RISVEC_BCD_REUSE_COLSUM) != 0,
                                         (flags & RISVEC_BCD_REUSE_SSUM) != 0, (flags & RISVEC_BCD_REUSE_IDX) != 0,
                                         (flags & RISVEC_BCD_NO_THETA) == 0, (hipStream_t)stream));
}

int risvec_theta_from_index(const RisVecState* s, risvec_stream_t stream) {
    const char* fn = "risvec_theta_from_index";
    if (int rc = check_common(fn, s, nullptr)) return rc;
    REQ_PTR(s->theta, "state.theta"); REQ_PTR(s->theta_idx, "state.theta_idx");
    if (s->control_bit != 3) return fail(RISVEC_ERR_UNSUPPORTED, "%s: candidate indices exist for control_bit = 3 only", fn);
    return finish(fn, risvec::launch_theta_from_index(*s, (hipStream_t)stream));
}

int risvec_theta_by_index_supported(int32_t n_veh, int32_t n_ris) {
    return risvec::theta_by_index_supported(n_veh, n_ris) ? 1 : 0;
}

int risvec_set_phase(const RisVecState* s, const float* angle, risvec_stream_t stream) {
    const char* fn = "risvec_set_phase";
    if (int rc = check_common(fn, s, nullptr)) return rc;
    REQ_PTR(angle, "angle"); REQ_PTR(s->theta, "state.theta");
    return finish(fn, risvec::launch_set_phase(*s, angle, (hipStream_t)stream));
}

int risvec_random_phase(const RisVecState* s, const int32_t* idx, uint64_t seed, uint32_t counter,
                        risvec_stream_t stream) {
    const char* fn = "risvec_random_phase";
    if (int rc = check_common(fn, s, nullptr)) return rc;
    OPT_PTR(idx, "idx"); REQ_PTR(s->theta, "state.theta");
    return finish(fn, risvec::launch_random_phase(*s, idx, seed, counter, (hipStream_t)stream));
}

int risvec_step(const RisVecState* s, const RisVecParams* p, const float* action, const int32_t* partner,
                const int32_t* n_groups, const int32_t* arrivals, uint64_t seed, uint32_t counter,
                uint32_t flags, risvec_stream_t stream) {
    const char* fn = "risvec_step";
    if (!p) return fail(RISVEC_ERR_ARG, "%s: params is NULL", fn);
    if (int rc = check_common(fn, s, p)) return rc;
    if (int rc = check_step(fn, s, action, partner, n_groups, arrivals, flags, false)) return rc;
    if (flags & RISVEC_STEP_THETA_BY_INDEX)
        return fail(RISVEC_ERR_ARG, "%s: RISVEC_STEP_THETA_BY_INDEX is a form of the fused entry points (this one reads no theta)", fn);
    return finish(fn, risvec::launch_step(*s, *p, action, partner, n_groups, arrivals, seed, counter, flags,
                                          false, (hipStream_t)stream));
}

int risvec_data_rate(const RisVecState* s, const RisVecParams* p, const float* p_off, const int32_t* partner,
                     const int32_t* n_groups, float* rate_out, risvec_stream_t stream) {
    const char* fn = "risvec_data_rate";
    if (!p) return fail(RISVEC_ERR_ARG, "%s: params is NULL", fn);
    if (int rc = check_common(fn, s, p)) return rc;
    REQ_PTR(p_off, "p_off"); REQ_PTR(partner, "partner"); REQ_PTR(n_groups, "n_groups");
    REQ_PTR(rate_out, "rate_out"); REQ_PTR(s->gain, "state.gain");
    return finish(fn, risvec::launch_data_rate(*s, *p, p_off, partner, n_groups, rate_out, (hipStream_t)stream));
}

int risvec_step_fused(const RisVecState* s, const RisVecParams* p, const float* action,
                      const int32_t* partner, const int32_t* n_groups, const int32_t* arrivals,
                      uint64_t seed, uint32_t counter, uint32_t flags, risvec_stream_t stream) {
    const char* fn = "risvec_step_fused";
    if (!p) return fail(RISVEC_ERR_ARG, "%s: params is NULL", fn);
    if (int rc = check_common(fn, s, p)) return rc;
    if (int rc = check_step(fn, s, action, partner, n_groups, arrivals, flags, true)) return rc;
    if (flags & RISVEC_STEP_THETA_BY_INDEX) {                  // theta as the last sweep's candidate indices (no sweep here)
        REQ_PTR(s->theta_idx, "state.theta_idx");
        if (s->control_bit != 3 || !risvec::theta_by_index_supported(s->n_veh, s->n_ris) || (flags & RISVEC_STEP_STEER))
            return fail(RISVEC_ERR_UNSUPPORTED, "%s: no theta-by-index form of the fused step for control_bit=%d, n_veh=%d, "
                        "n_ris=%d%s", fn, s->control_bit, s->n_veh, s->n_ris, (flags & RISVEC_STEP_STEER) ? " with RISVEC_STEP_STEER" : "");
    }
    return finish(fn, risvec::launch_step(*s, *p, action, partner, n_groups, arrivals, seed, counter, flags,
                                          true, (hipStream_t)stream));
}

int risvec_step_fused_multi(const RisVecState* s, const RisVecParams* p, int32_t n_steps, const float* actions,
                            const int32_t* partner, const int32_t* n_groups, const int32_t* arrivals, uint64_t seed,
                            uint32_t counter, uint32_t flags, const RisVecTraj* traj, risvec_stream_t stream) {
    const char* fn = "risvec_step_fused_multi";
    if (!p) return fail(RISVEC_ERR_ARG, "%s: params is NULL", fn);
    if (int rc = check_common(fn, s, p)) return rc;
    if (n_steps < 1 || n_steps > (1 << 20)) return fail(RISVEC_ERR_ARG, "%s: n_steps=%d outside [1, 2^20]", fn, n_steps);
    if (flags & (RISVEC_STEP_REUSE_COLSUM | RISVEC_STEP_REUSE_SSUM | RISVEC_STEP_REUSE_IDX | RISVEC_STEP_STEER | RISVEC_STEP_THETA_BY_INDEX))
        return fail(RISVEC_ERR_ARG, "%s: the BCD / steering flags (0x%x) are not accepted by the multi-step launch", fn, flags);
    if (int rc = check_step(fn, s, actions, partner, n_groups, arrivals, flags, true)) return rc;
    if (traj) { OPT_PTR(traj->reward, "traj.reward"); OPT_PTR(traj->obs, "traj.obs"); OPT_PTR(traj->metrics, "traj.metrics"); }
    hipStream_t st = (hipStream_t)stream;
    const risvec::StepArgs a = risvec::make_step_args(*s, actions, partner, n_groups, arrivals, seed, counter, flags);
    const hipError_t err = risvec::launch_step_fused_multi(*s, *p, a, n_steps, traj, st);
    if (err != hipErrorNotSupported) return finish(fn, err);
    // Shapes without a compile-time fused kernel: step 0 through the single fused launch (gains computed and stored, its
    // records copied out), steps 1 .. T-1 in ONE launch on those stored gains (h_r / theta cannot change inside the call,
    // so they are the gains T fused launches would recompute): bit-identical to T risvec_step_fused calls.
    const long long ev = (long long)s->n_envs * s->n_veh;
    const hipError_t e1 = risvec::launch_step(*s, *p, actions, partner, n_groups, arrivals, seed, counter, flags, true, st);
    if (e1 != hipSuccess) return finish(fn, e1);
    if (traj) {
        hipError_t e2 = hipSuccess;
        if (traj->reward) e2 = hipMemcpyAsync(traj->reward, s->reward, ev * 4, hipMemcpyDeviceToDevice, st);
        if (e2 == hipSuccess && traj->obs && (flags & RISVEC_STEP_OBS))
            e2 = hipMemcpyAsync(traj->obs, s->obs, ev * 20, hipMemcpyDeviceToDevice, st);
        if (e2 == hipSuccess && traj->metrics)
            e2 = hipMemcpyAsync(traj->metrics, s->metrics, (size_t)s->n_envs * RISVEC_METRICS * 4, hipMemcpyDeviceToDevice, st);
        if (e2 != hipSuccess) return finish(fn, e2);
    }
    if (n_steps == 1) return RISVEC_OK;
    const risvec::StepArgs rest = risvec::make_step_args(*s, actions + 2 * ev, partner, n_groups,
                                                         arrivals ? arrivals + ev : nullptr, seed, counter + 1u, flags);
    RisVecTraj tj{nullptr, nullptr, nullptr};
    if (traj) {
        tj.reward = traj->reward ? traj->reward + ev : nullptr;
        tj.obs = (traj->obs && (flags & RISVEC_STEP_OBS)) ? traj->obs + ev * 5 : nullptr;
        tj.metrics = traj->metrics ? traj->metrics + (long long)s->n_envs * RISVEC_METRICS : nullptr;
    }
    return finish(fn, risvec::launch_step_multi(*s, *p, rest, n_steps - 1, &tj, st));
}

int risvec_step_multi(const RisVecState* s, const RisVecParams* p, int32_t n_steps, const float* actions,
                      const int32_t* partner, const int32_t* n_groups, const int32_t* arrivals, uint64_t seed,
                      uint32_t counter, uint32_t flags, const RisVecTraj* traj, risvec_stream_t stream) {
    const char* fn = "risvec_step_multi";
    if (!p) return fail(RISVEC_ERR_ARG, "%s: params is NULL", fn);
    if (int rc = check_common(fn, s, p)) return rc;
    if (n_steps < 1 || n_steps > (1 << 20)) return fail(RISVEC_ERR_ARG, "%s: n_steps=%d outside [1, 2^20]", fn, n_steps);
    if (flags & (RISVEC_STEP_REUSE_COLSUM | RISVEC_STEP_REUSE_SSUM | RISVEC_STEP_REUSE_IDX | RISVEC_STEP_STEER | RISVEC_STEP_THETA_BY_INDEX))
        return fail(RISVEC_ERR_ARG, "%s: the BCD / steering flags (0x%x) are not accepted by the multi-step launch", fn, flags);
    if (int rc = check_step(fn, s, actions, partner, n_groups, arrivals, flags, false)) return rc;
    if (traj) { OPT_PTR(traj->reward, "traj.reward"); OPT_PTR(traj->obs, "traj.obs"); OPT_PTR(traj->metrics, "traj.metrics"); }
    const risvec::StepArgs a = risvec::make_step_args(*s, actions, partner, n_groups, arrivals, seed, counter, flags);
    RisVecTraj tj{nullptr, nullptr, nullptr};
    if (traj) {
        tj = *traj;
        if (!(flags & RISVEC_STEP_OBS)) tj.obs = nullptr;      // obs records need the obs flag, as in the fused form
    }
    return finish(fn, risvec::launch_step_multi(*s, *p, a, n_steps, &tj, (hipStream_t)stream));
}

int risvec_sarl_step(const RisVecState* s, const RisVecSarlParams* p, const float* action_power,
                     const float* action_phase, const int32_t* arrivals, uint64_t seed, uint32_t counter,
                     uint32_t flags, risvec_stream_t stream) {
    const char* fn = "risvec_sarl_step";
    if (!p) return fail(RISVEC_ERR_ARG, "%s: params is NULL", fn);
    if (p->abi_version != RISVEC_ABI_VERSION || p->struct_bytes != sizeof(RisVecSarlParams))
        return fail(RISVEC_ERR_ARG, "%s: RisVecSarlParams ABI mismatch (version %u/%u, bytes %u/%zu)", fn,
                    p->abi_version, (unsigned)RISVEC_ABI_VERSION, p->struct_bytes, sizeof(RisVecSarlParams));
    if (int rc = check_common(fn, s, nullptr)) return rc;
    REQ_PTR(action_power, "action_power"); OPT_PTR(action_phase, "action_phase"); OPT_PTR(arrivals, "arrivals");
    REQ_PTR(s->h_r, "state.h_r"); REQ_PTR(s->theta, "state.theta"); REQ_PTR(s->b, "state.b"); REQ_PTR(s->pl, "state.pl");
    REQ_PTR(s->gain, "state.gain"); REQ_PTR(s->data_buf, "state.data_buf"); REQ_PTR(s->rate, "state.rate");
    REQ_PTR(s->data_t, "state.data_t"); REQ_PTR(s->data_p, "state.data_p"); REQ_PTR(s->reward, "state.reward");
    REQ_PTR(s->over_power, "state.over_power"); REQ_PTR(s->over_data, "state.over_data");
    REQ_PTR(s->metrics, "state.metrics");
    if (flags & RISVEC_STEP_OBS) REQ_PTR(s->obs, "state.obs");
    if (flags & ~(uint32_t)RISVEC_STEP_OBS) return fail(RISVEC_ERR_ARG, "%s: unknown flag bits 0x%x", fn, flags);
    return finish(fn, risvec::launch_sarl_step(*s, *p, action_power, action_phase, arrivals, seed, counter, flags,
                                               (hipStream_t)stream));
}

int risvec_step_fused_bcd(const RisVecState* s, const RisVecParams* p, const float* action,
                          const int32_t* partner, const int32_t* n_groups, const int32_t* arrivals,
                          uint64_t seed, uint32_t counter, uint32_t flags, risvec_stream_t stream) {
    const char* fn = "risvec_step_fused_bcd";
    if (!p) return fail(RISVEC_ERR_ARG, "%s: params is NULL", fn);
    if (int rc = check_common(fn, s, p)) return rc;
    if (int rc = check_step(fn, s, action, partner, n_groups, arrivals, flags, true)) return rc;
    REQ_PTR(s->c_col, "state.c_col");
    OPT_PTR(s->theta_idx, "state.theta_idx");
    if ((flags & RISVEC_STEP_REUSE_IDX) && !s->theta_idx)
        return fail(RISVEC_ERR_ARG, "%s: RISVEC_STEP_REUSE_IDX needs state.theta_idx", fn);
    if (flags & RISVEC_STEP_THETA_BY_INDEX) {
        if (!(flags & RISVEC_STEP_REUSE_IDX) || s->control_bit != 3)
            return fail(RISVEC_ERR_ARG, "%s: RISVEC_STEP_THETA_BY_INDEX needs RISVEC_STEP_REUSE_IDX and control_bit = 3", fn);
        if (!risvec::theta_by_index_supported(s->n_veh, s->n_ris))
            return fail(RISVEC_ERR_UNSUPPORTED, "%s: no theta-by-index form of the fused step at n_veh=%d, n_ris=%d", fn,
                        s->n_veh, s->n_ris);
        if (flags & RISVEC_STEP_STEER)
            return fail(RISVEC_ERR_ARG, "%s: RISVEC_STEP_THETA_BY_INDEX and RISVEC_STEP_STEER exclude each other", fn);
    }
    return finish(fn, risvec::launch_step_fused_bcd(*s, *p, action, partner, n_groups, arrivals, seed,
                                                    counter, flags, (hipStream_t)stream));
}

int risvec_step_ring(const RisVecState* s, const RisVecParams* p, const RisVecStepRing* ring, const float* action,
                     const int32_t* partner, const int32_t* n_groups, const int32_t* arrivals, uint64_t seed,
                     uint32_t counter, uint32_t flags, int32_t fused, risvec_stream_t stream) {
    const char* fn = "risvec_step_ring";
    if (!p) return fail(RISVEC_ERR_ARG, "%s: params is NULL", fn);
    if (!ring) return fail(RISVEC_ERR_ARG, "%s: ring is NULL", fn);
    if (int rc = check_common(fn, s, p)) return rc;
    if (int rc = check_step(fn, s, action, partner, n_groups, arrivals, flags, fused != 0)) return rc;
    const uint32_t need = RISVEC_STEP_POLICY_ACTION | RISVEC_STEP_OBS;
    if ((flags & need) != need)
        return fail(RISVEC_ERR_ARG, "%s: flags must hold RISVEC_STEP_POLICY_ACTION | RISVEC_STEP_OBS (the ring stores the raw policy "
                    "output and both observations)", fn);
    if (flags & (RISVEC_STEP_STEER | RISVEC_STEP_THETA_BY_INDEX | RISVEC_STEP_REUSE_COLSUM | RISVEC_STEP_REUSE_SSUM | RISVEC_STEP_REUSE_IDX))
        return fail(RISVEC_ERR_ARG, "%s: steering / theta-by-index / BCD flags (0x%x) are not accepted here", fn, flags);
    const RisVecReplay& rb = ring->rb;
    const int V = s->n_veh;
    if (V != 4 && V != 8 && V != 16)
        return fail(RISVEC_ERR_UNSUPPORTED, "%s: n_veh=%d (the fused transition store exists for 4, 8 and 16 vehicles)", fn, V);
    if (rb.n_agents != V || rb.input_shape != 5 || rb.n_actions != V + 2)
        return fail(RISVEC_ERR_SHAPE, "%s: the ring must have n_agents = n_veh = %d, input_shape = 5, n_actions = %d (got %d, %d, %d)",
                    fn, V, V + 2, rb.n_agents, rb.input_shape, rb.n_actions);
    if (rb.mem_size < s->n_envs) return fail(RISVEC_ERR_SHAPE, "%s: n_envs=%d transitions do not fit mem_size=%lld", fn, s->n_envs,
                                             (long long)rb.mem_size);
    if (ring->mem_cntr < 0) return fail(RISVEC_ERR_ARG, "%s: mem_cntr < 0", fn);
    REQ_PTR(rb.state_memory, "ring.state_memory"); REQ_PTR(rb.action_memory, "ring.action_memory");
    REQ_PTR(rb.reward_global_memory, "ring.reward_global_memory"); REQ_PTR(rb.reward_local_memory, "ring.reward_local_memory");
    REQ_PTR(rb.new_state_memory, "ring.new_state_memory"); REQ_PTR(rb.terminal_memory, "ring.terminal_memory");
    REQ_PTR(rb.mask_memory, "ring.mask_memory"); REQ_PTR(ring->probs, "ring.probs"); OPT_PTR(ring->mask, "ring.mask");
    REQ_PTR(s->obs, "state.obs");
    risvec::StepRing r{rb.state_memory, rb.action_memory, rb.reward_global_memory, rb.reward_local_memory, rb.new_state_memory,
                       rb.terminal_memory, rb.mask_memory, ring->probs, ring->mask, (long long)(ring->mem_cntr % rb.mem_size),
                       (long long)rb.mem_size, ring->done ? 1 : 0};
    const hipError_t err = risvec::launch_step(*s, *p, action, partner, n_groups, arrivals, seed, counter, flags, fused != 0,
                                               (hipStream_t)stream, &r);
    if (err == hipErrorNotSupported)
        return fail(RISVEC_ERR_UNSUPPORTED, "%s: no fused-gains kernel with the transition store at n_veh=%d, n_ris=%d (use "
                    "risvec_step_fused + risvec_replay_store_policy)", fn, V, s->n_ris);
    return finish(fn, err);
}

// ---------------------------------------------------------------------------------------------
// NOMA grouping stage (f2)
// ---------------------------------------------------------------------------------------------
void risvec_noma_default_params(RisVecNomaParams* p, int32_t n_veh) {
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->min_pair_target = n_veh / 4 > 1 ? n_veh / 4 : 1;               // TRAIN:489
    p->mwm_backoff_rounds = 5; p->mwm_allow_singles = 1;               // TRAIN:440, 436
    p->qos_enable = 0; p->relax_topk_step = 1;                         // TRAIN:750 (Config default), 728
    p->freeze_group_in_episode = 1; p->freeze_recalc_every = 0;        // TRAIN:738-739
    p->mask_enable = 1;                                                // TRAIN:498
    p->mwm_accept_quantile = 0.10; p->mwm_accept_q_step = 0.05;        // TRAIN:439, 441
    p->completion_min_quantile = 0.30;                                 // TRAIN:282
    p->score_w_delta_db = 1.0; p->score_w_history = 0.3f;              // TRAIN:716-717
    p->abs_gain_min_db = -HUGE_VAL;                                    // TRAIN:723
    p->qos_soft_penalty = 6.0; p->qos_R_min = 0.0;                     // TRAIN:172, 751
    p->noise_power = std::pow(10.0, -174.0 / 10.0) / 1000.0 * 1.0e6;   // ENV:72-76
    p->P_max = 1.0;                                                    // ENV:125
    p->relax_tau_factor = 0.95; p->tau_back_floor_db = 3.0;            // TRAIN:729, 1499
    p->freeze_reward_drop_ratio = 0.05; p->freeze_unstick_prob = 0.0;  // TRAIN:741, 740
    p->pair_hist_decay = 0.97f;                                        // TRAIN:719
}

int64_t risvec_noma_scratch_bytes(int32_t n_envs, int32_t n_veh) {
    if (n_envs < 1 || n_veh < 1 || n_veh > RISVEC_NOMA_MAX_VEH) return 0;
    return risvec::noma_scratch_bytes(n_envs, n_veh);
}

static int check_noma(const char* fn, const RisVecNomaState* ns) {
    if (!ns) return fail(RISVEC_ERR_ARG, "%s: noma state is NULL", fn);
    if (ns->n_envs < 1) return fail(RISVEC_ERR_SHAPE, "%s: n_envs=%d must be >= 1", fn, ns->n_envs);
    if (ns->n_veh < 1 || ns->n_veh > RISVEC_NOMA_MAX_VEH)
        return fail(RISVEC_ERR_SHAPE, "%s: n_veh=%d outside [1,%d]", fn, ns->n_veh, RISVEC_NOMA_MAX_VEH);
    if (ns->env_offset < 0 || ns->env_offset + ns->n_envs > 0xFFFFFFFFLL)
        return fail(RISVEC_ERR_SHAPE, "%s: env_offset+n_envs must fit 32 bits", fn);
    return RISVEC_OK;
}

int risvec_noma_begin_episode(const RisVecNomaState* ns, risvec_stream_t stream) {
    const char* fn = "risvec_noma_begin_episode";
    if (int rc = check_noma(fn, ns)) return rc;
    REQ_PTR(ns->hist, "noma.hist"); REQ_PTR(ns->streak, "noma.streak"); REQ_PTR(ns->flags, "noma.flags");
    REQ_PTR(ns->pending, "noma.pending");
    return finish(fn, risvec::launch_noma_begin_episode(*ns, (hipStream_t)stream));
}

int risvec_noma_mask(const RisVecNomaState* ns, const float* gain, const double* gdb15, double q_now, int32_t K_now,
                     risvec_stream_t stream) {
    const char* fn = "risvec_noma_mask";
    if (int rc = check_noma(fn, ns)) return rc;
    OPT_PTR(gdb15, "gdb15");
    if (!gdb15) REQ_PTR(gain, "gain");
    REQ_PTR(ns->tau, "noma.tau");
    if (K_now >= 1) REQ_PTR(ns->mask, "noma.mask");
    if (!(q_now >= 0.0 && q_now <= 1.0)) return fail(RISVEC_ERR_ARG, "%s: q_now=%g outside [0,1]", fn, q_now);
    return finish(fn, risvec::launch_noma_mask(*ns, gain, gdb15, q_now, K_now, (hipStream_t)stream));
}

static int check_noma_state(const char* fn, const RisVecNomaState* ns) {
    REQ_PTR(ns->hist, "noma.hist"); REQ_PTR(ns->streak, "noma.streak"); REQ_PTR(ns->partner, "noma.partner");
    REQ_PTR(ns->n_groups, "noma.n_groups"); REQ_PTR(ns->last_global, "noma.last_global");
    REQ_PTR(ns->best_global, "noma.best_global"); REQ_PTR(ns->flags, "noma.flags");
    REQ_PTR(ns->pending, "noma.pending");
    return RISVEC_OK;
}

static int noma_group_impl(const char* fn, const RisVecNomaState* ns, const RisVecNomaParams* np, const float* gain,
                           const double* gdb12, const float* p_off01, int p01_raw, int32_t use_mask, int32_t K_back,
                           const double* tau_back, const float* prev_global, int32_t prev_global_stride, int32_t i_step,
                           const float* u_unstick, uint64_t seed, uint32_t counter, int32_t* info_out,
                           risvec_stream_t stream) {
    if (int rc = check_noma(fn, ns)) return rc;
    if (!np) return fail(RISVEC_ERR_ARG, "%s: params is NULL", fn);
    REQ_PTR(gain, "gain"); OPT_PTR(gdb12, "gdb12"); OPT_PTR(p_off01, "p_off01"); REQ_PTR(tau_back, "tau_back");
    OPT_PTR(u_unstick, "u_unstick"); OPT_PTR(info_out, "info_out");
    if (int rc = check_noma_state(fn, ns)) return rc;
    if (np->qos_enable && !p_off01) return fail(RISVEC_ERR_ARG, "%s: qos_enable needs p_off01", fn);
    if (np->mask_enable && use_mask) REQ_PTR(ns->mask, "noma.mask");
    if (prev_global && prev_global_stride < 1)
        return fail(RISVEC_ERR_ARG, "%s: prev_global_stride=%d must be >= 1", fn, prev_global_stride);
    if (prev_global && (reinterpret_cast<uintptr_t>(prev_global) & 3u))
        return fail(RISVEC_ERR_ARG, "%s: prev_global is not 4-byte aligned", fn);
    if (K_back < 0) return fail(RISVEC_ERR_ARG, "%s: K_back=%d must be >= 0", fn, K_back);
    const long long need = risvec::noma_scratch_bytes(ns->n_envs, ns->n_veh);
    if (need > 0 && (!ns->scratch || ns->scratch_bytes < need))
        return fail(RISVEC_ERR_ARG, "%s: noma.scratch holds %lld bytes, risvec_noma_scratch_bytes(%d, %d) = %lld", fn,
                    ns->scratch ? (long long)ns->scratch_bytes : 0LL, ns->n_envs, ns->n_veh, need);
    if (need > 0) REQ_PTR(ns->scratch, "noma.scratch");
    if (np->mwm_backoff_rounds < 0 || np->mwm_backoff_rounds > 64)
        return fail(RISVEC_ERR_ARG, "%s: mwm_backoff_rounds=%d outside [0,64]", fn, np->mwm_backoff_rounds);
    return finish(fn, risvec::launch_noma_group(*ns, *np, gain, gdb12, p_off01, p01_raw, use_mask, K_back, tau_back,
                                                prev_global, prev_global_stride, i_step, u_unstick, seed, counter,
                                                info_out, (hipStream_t)stream));
}

int risvec_noma_group(const RisVecNomaState* ns, const RisVecNomaParams* np, const float* gain, const double* gdb12,
                      const float* p_off01, int32_t use_mask, int32_t K_back, const double* tau_back,
                      const float* prev_global, int32_t prev_global_stride, int32_t i_step, const float* u_unstick,
                      uint64_t seed, uint32_t counter, int32_t* info_out, risvec_stream_t stream) {
    return noma_group_impl("risvec_noma_group", ns, np, gain, gdb12, p_off01, 0, use_mask, K_back, tau_back, prev_global,
                           prev_global_stride, i_step, u_unstick, seed, counter, info_out, stream);
}

int risvec_noma_group_raw(const RisVecNomaState* ns, const RisVecNomaParams* np, const float* gain, const double* gdb12,
                          const float* power_raw, int32_t use_mask, int32_t K_back, const double* tau_back,
                          const float* prev_global, int32_t prev_global_stride, int32_t i_step, const float* u_unstick,
                          uint64_t seed, uint32_t counter, int32_t* info_out, risvec_stream_t stream) {
    return noma_group_impl("risvec_noma_group_raw", ns, np, gain, gdb12, power_raw, 1, use_mask, K_back, tau_back,
                           prev_global, prev_global_stride, i_step, u_unstick, seed, counter, info_out, stream);
}

int risvec_noma_flush(const RisVecNomaState* ns, const RisVecNomaParams* np, risvec_stream_t stream) {
    const char* fn = "risvec_noma_flush";
    if (int rc = check_noma(fn, ns)) return rc;
    if (!np) return fail(RISVEC_ERR_ARG, "%s: params is NULL", fn);
    if (int rc = check_noma_state(fn, ns)) return rc;
    return finish(fn, risvec::launch_noma_flush(*ns, np->pair_hist_decay, (hipStream_t)stream));
}

// ---------------------------------------------------------------------------------------------
// replay ring buffer + marshalling (f3)
// ---------------------------------------------------------------------------------------------
static int check_replay(const char* fn, const RisVecReplay* rb) {
    if (!rb) return fail(RISVEC_ERR_ARG, "%s: replay is NULL", fn);
    if (rb->n_agents < 1 || rb->n_agents > RISVEC_MAX_VEH || rb->input_shape < 1 || rb->n_actions < 1)
        return fail(RISVEC_ERR_SHAPE, "%s: n_agents=%d input_shape=%d n_actions=%d", fn, rb->n_agents, rb->input_shape,
                    rb->n_actions);
    if (rb->mem_size < 1) return fail(RISVEC_ERR_SHAPE, "%s: mem_size=%lld must be >= 1", fn, (long long)rb->mem_size);
    REQ_PTR(rb->state_memory, "replay.state_memory"); REQ_PTR(rb->action_memory, "replay.action_memory");
    REQ_PTR(rb->reward_global_memory, "replay.reward_global_memory");
    REQ_PTR(rb->reward_local_memory, "replay.reward_local_memory");
    REQ_PTR(rb->new_state_memory, "replay.new_state_memory"); REQ_PTR(rb->terminal_memory, "replay.terminal_memory");
    REQ_PTR(rb->mask_memory, "replay.mask_memory");
    return RISVEC_OK;
}

static int replay_store_impl(const char* fn, const RisVecReplay* rb, int64_t mem_cntr, int32_t n, const float* state,
                             const float* action, const float* power_raw, const float* probs, const float* reward_g,
                             int32_t reward_g_stride, const float* reward_l, const float* state_, const uint8_t* done,
                             int32_t done_all, const uint8_t* mask, float* state_carry, risvec_stream_t stream) {
    if (int rc = check_replay(fn, rb)) return rc;
    if (n < 1 || n > rb->mem_size)
        return fail(RISVEC_ERR_SHAPE, "%s: n=%d outside [1, mem_size=%lld] (a batch may not overwrite itself)", fn, n,
                    (long long)rb->mem_size);
    if (mem_cntr < 0) return fail(RISVEC_ERR_ARG, "%s: mem_cntr=%lld must be >= 0", fn, (long long)mem_cntr);
    if (reward_g_stride < 1) return fail(RISVEC_ERR_ARG, "%s: reward_g_stride=%d must be >= 1", fn, reward_g_stride);
    REQ_PTR(state, "state"); REQ_PTR(reward_l, "reward_l"); REQ_PTR(state_, "state_");
    if (action) {
        REQ_PTR(action, "action");
    } else {
        REQ_PTR(power_raw, "power_raw"); REQ_PTR(probs, "probs");
        if (rb->n_actions != rb->n_agents + 2)
            return fail(RISVEC_ERR_SHAPE, "%s: the policy-output form needs n_actions = n_agents + 2 (got %d, %d)", fn,
                        rb->n_actions, rb->n_agents);
    }
    OPT_PTR(state_carry, "state_carry"); OPT_PTR(mask, "mask");
    if (state_carry == state || state_carry == state_)
        return fail(RISVEC_ERR_ARG, "%s: state_carry must not alias state / state_ (other lanes are reading them)", fn);
    if (!reward_g || (reinterpret_cast<uintptr_t>(reward_g) & 3u))
        return fail(RISVEC_ERR_ARG, "%s: reward_g is NULL or not 4-byte aligned", fn);
    return finish(fn, risvec::launch_replay_store(*rb, mem_cntr, n, state, action, power_raw, probs, reward_g, reward_g_stride, reward_l,
                                                  state_, done, done_all, mask, state_carry, (hipStream_t)stream));
}

int risvec_replay_store(const RisVecReplay* rb, int64_t mem_cntr, int32_t n, const float* state, const float* action,
                        const float* reward_g, int32_t reward_g_stride, const float* reward_l, const float* state_,
                        const uint8_t* done, int32_t done_all, const uint8_t* mask, float* state_carry,
                        risvec_stream_t stream) {
    if (!action && rb && rb->mem_size >= 1 && rb->n_agents >= 1 && rb->state_memory)   // (a malformed rb is reported first)
        return fail(RISVEC_ERR_ARG, "risvec_replay_store: action is NULL");
    return replay_store_impl("risvec_replay_store", rb, mem_cntr, n, state, action, nullptr, nullptr, reward_g, reward_g_stride,
                             reward_l, state_, done, done_all, mask, state_carry, stream);
}

int risvec_replay_store_policy(const RisVecReplay* rb, int64_t mem_cntr, int32_t n, const float* state,
                               const float* power_raw, const float* probs, const float* reward_g, int32_t reward_g_stride,
                               const float* reward_l, const float* state_, const uint8_t* done, int32_t done_all,
                               const uint8_t* mask, float* state_carry, risvec_stream_t stream) {
    return replay_store_impl("risvec_replay_store_policy", rb, mem_cntr, n, state, nullptr, power_raw, probs, reward_g,
                             reward_g_stride, reward_l, state_, done, done_all, mask, state_carry, stream);
}

int risvec_replay_sample(const RisVecReplay* rb, int64_t max_mem, int32_t batch, const int64_t* idx, uint64_t seed,
                         uint32_t counter, float* states, float* actions, float* rewards_g, float* rewards_l,
                         float* states_, uint8_t* dones, float* masks, int64_t* idx_out, risvec_stream_t stream) {
    const char* fn = "risvec_replay_sample";
    if (int rc = check_replay(fn, rb)) return rc;
    if (batch < 1) return fail(RISVEC_ERR_SHAPE, "%s: batch=%d must be >= 1", fn, batch);
    if (max_mem < 1 || max_mem > rb->mem_size)
        return fail(RISVEC_ERR_ARG, "%s: max_mem=%lld outside [1, mem_size=%lld] (sampling an empty buffer?)", fn,
                    (long long)max_mem, (long long)rb->mem_size);
    OPT_PTR(idx, "idx"); OPT_PTR(idx_out, "idx_out");
    REQ_PTR(states, "states"); REQ_PTR(actions, "actions"); REQ_PTR(rewards_g, "rewards_g");
    REQ_PTR(rewards_l, "rewards_l"); REQ_PTR(states_, "states_"); REQ_PTR(dones, "dones"); REQ_PTR(masks, "masks");
    return finish(fn, risvec::launch_replay_sample(*rb, max_mem, batch, idx, seed, counter, states, actions, rewards_g,
                                                   rewards_l, states_, dones, masks, idx_out, (hipStream_t)stream));
}

int risvec_marshal_actions(int32_t n_envs, int32_t n_veh, const float* power_raw, const float* probs,
                           float cpu_share_floor, float* action_env, float* p_off01, float* action_store,
                           risvec_stream_t stream) {
    const char* fn = "risvec_marshal_actions";
    if (n_envs < 1) return fail(RISVEC_ERR_SHAPE, "%s: n_envs=%d must be >= 1", fn, n_envs);
    if (n_veh < 1 || n_veh > RISVEC_MAX_VEH)
        return fail(RISVEC_ERR_SHAPE, "%s: n_veh=%d outside [1,%d]", fn, n_veh, RISVEC_MAX_VEH);
    REQ_PTR(power_raw, "power_raw"); OPT_PTR(probs, "probs"); OPT_PTR(action_env, "action_env");
    OPT_PTR(p_off01, "p_off01"); OPT_PTR(action_store, "action_store");
    if (action_store && !probs) return fail(RISVEC_ERR_ARG, "%s: action_store needs probs", fn);
    if ((long long)n_envs * n_veh * (n_veh + 2) >= (1LL << 31))
        return fail(RISVEC_ERR_SHAPE, "%s: n_envs*n_veh*(n_veh+2) must stay below 2^31", fn);
    const float fl = cpu_share_floor < 0.0f ? 0.0f : (cpu_share_floor > 0.95f ? 0.95f : cpu_share_floor);
    return finish(fn, risvec::launch_marshal_actions(n_envs, n_veh, power_raw, probs, fl, action_env, p_off01,
                                                     action_store, (hipStream_t)stream));
}

int risvec_policy_sample(int32_t n_envs, int32_t n_veh, int64_t env_offset, const float* heads, const uint8_t* mask,
                         const float* tau, const uint8_t* hard, const float* eps, const float* expo, uint64_t seed,
                         uint32_t counter, float cpu_share_floor, float* power_raw, float* probs, float* onehot,
                         float* action_env, float* p_off01, float* action_store, risvec_stream_t stream) {
    const char* fn = "risvec_policy_sample";
    if (n_envs < 1) return fail(RISVEC_ERR_SHAPE, "%s: n_envs=%d must be >= 1", fn, n_envs);
    if (n_veh < 1 || n_veh > RISVEC_MAX_VEH)
        return fail(RISVEC_ERR_SHAPE, "%s: n_veh=%d outside [1,%d]", fn, n_veh, RISVEC_MAX_VEH);
    if (env_offset < 0 || env_offset + n_envs > 0xFFFFFFFFLL)
        return fail(RISVEC_ERR_SHAPE, "%s: env_offset+n_envs must fit 32 bits", fn);
    REQ_PTR(heads, "heads"); REQ_PTR(tau, "tau"); REQ_PTR(power_raw, "power_raw"); REQ_PTR(probs, "probs");
    OPT_PTR(mask, "mask"); OPT_PTR(eps, "eps"); OPT_PTR(expo, "expo"); OPT_PTR(onehot, "onehot");
    OPT_PTR(action_env, "action_env"); OPT_PTR(p_off01, "p_off01"); OPT_PTR(action_store, "action_store");
    const float fl = cpu_share_floor < 0.0f ? 0.0f : (cpu_share_floor > 0.95f ? 0.95f : cpu_share_floor);
    return finish(fn, risvec::launch_policy_sample(n_envs, n_veh, env_offset, heads, mask, tau, hard, eps, expo, seed, counter,
                                                   fl, power_raw, probs, onehot, action_env, p_off01, action_store,
                                                   (hipStream_t)stream));
}

int risvec_policy_layer1(int32_t n_envs, int32_t n_veh, int32_t in_dims, int32_t f1, const float* obs, const float* W1,
                         const float* b1, const float* ln_w, const float* ln_b, float* out, risvec_stream_t stream) {
    const char* fn = "risvec_policy_layer1";
    if (n_envs < 1 || n_veh < 1 || n_veh > 65535) return fail(RISVEC_ERR_SHAPE, "%s: n_envs=%d n_veh=%d", fn, n_envs, n_veh);
    if (in_dims < 1 || f1 < 1 || f1 > 1024 || (long long)(in_dims + 3) * f1 * 4 > 64 * 1024)
        return fail(RISVEC_ERR_SHAPE, "%s: in_dims=%d f1=%d (f1 <= 1024, (in+3)*f1 floats must fit 64 KB of LDS)", fn,
                    in_dims, f1);
    REQ_PTR(obs, "obs"); REQ_PTR(W1, "W1"); REQ_PTR(b1, "b1"); REQ_PTR(ln_w, "ln_w"); REQ_PTR(ln_b, "ln_b"); REQ_PTR(out, "out");
    return finish(fn, risvec::launch_policy_layer1(n_envs, n_veh, in_dims, f1, obs, W1, b1, ln_w, ln_b, out,
                                                   (hipStream_t)stream));
}

int risvec_policy_layer1_split16(int32_t n_envs, int32_t n_veh, int32_t in_dims, int32_t f1, const float* obs,
                                 const float* W1, const float* b1, const float* ln_w, const float* ln_b, void* out16,
                                 risvec_stream_t stream) {
    const char* fn = "risvec_policy_layer1_split16";
    if (n_envs < 1 || n_veh < 1 || n_veh > 65535) return fail(RISVEC_ERR_SHAPE, "%s: n_envs=%d n_veh=%d", fn, n_envs, n_veh);
    if (in_dims < 1 || in_dims > 8 || f1 < 4 || f1 > 1024 || f1 % 4 != 0 || (long long)(in_dims + 3) * f1 * 4 > 64 * 1024)
        return fail(RISVEC_ERR_SHAPE, "%s: in_dims=%d f1=%d (in_dims <= 8, f1 a multiple of 4 and <= 1024)", fn, in_dims, f1);
    REQ_PTR(obs, "obs"); REQ_PTR(W1, "W1"); REQ_PTR(b1, "b1"); REQ_PTR(ln_w, "ln_w"); REQ_PTR(ln_b, "ln_b"); REQ_PTR(out16, "out16");
    return finish(fn, risvec::launch_policy_layer1_split16(n_envs, n_veh, in_dims, f1, obs, W1, b1, ln_w, ln_b, out16,
                                                           (hipStream_t)stream));
}

int risvec_policy_mlp_supported(int32_t in_dims, int32_t f1, int32_t f2, int32_t n_heads) {
    return risvec::policy_mlp_supported(in_dims, f1, f2, n_heads) ? 1 : 0;
}

int risvec_policy_mlp(int32_t n_envs, int32_t n_veh, int32_t in_dims, int32_t f1, int32_t f2, int32_t n_heads, const float* obs,
                      const float* G, const void* W1F, const void* W2f, const float* w_unscale, const float* b2,
                      const float* ln2_w, const float* ln2_b, const void* WhF, const float* wh_unscale, const float* bh,
                      float* heads, risvec_stream_t stream) {
    const char* fn = "risvec_policy_mlp";
    if (n_envs < 1 || n_veh < 1 || n_veh > 65535) return fail(RISVEC_ERR_SHAPE, "%s: n_envs=%d n_veh=%d", fn, n_envs, n_veh);
    if (!risvec::policy_mlp_supported(in_dims, f1, f2, n_heads))
        return fail(RISVEC_ERR_UNSUPPORTED, "%s: in_dims=%d f1=%d f2=%d n_heads=%d (built for in_dims <= 5, f1 a multiple of 32 "
                    "and <= 1024, f2 = 128 or 256, n_heads <= 24; use risvec_policy_layer1 + a GEMM + risvec_policy_heads)", fn,
                    in_dims, f1, f2, n_heads);
    REQ_PTR(obs, "obs"); REQ_PTR(G, "G"); REQ_PTR(W1F, "W1F"); REQ_PTR(W2f, "W2f"); REQ_PTR(w_unscale, "w_unscale");
    REQ_PTR(b2, "b2"); REQ_PTR(ln2_w, "ln2_w"); REQ_PTR(ln2_b, "ln2_b"); REQ_PTR(WhF, "WhF"); REQ_PTR(wh_unscale, "wh_unscale");
    REQ_PTR(bh, "bh"); REQ_PTR(heads, "heads");
    return finish(fn, risvec::launch_policy_mlp(n_envs, n_veh, in_dims, f1, f2, n_heads, obs, G, W1F, W2f, w_unscale, b2, ln2_w,
                                                ln2_b, WhF, wh_unscale, bh, heads, (hipStream_t)stream));
}

int risvec_policy_heads(int32_t n_envs, int32_t n_veh, int32_t f2, int32_t n_heads, const float* g, const float* b2,
                        const float* ln_w, const float* ln_b, const float* Wh, const float* bh, float* heads,
                        risvec_stream_t stream) {
    const char* fn = "risvec_policy_heads";
    if (n_envs < 1 || n_veh < 1 || n_veh > 65535) return fail(RISVEC_ERR_SHAPE, "%s: n_envs=%d n_veh=%d", fn, n_envs, n_veh);
    if (f2 < 1 || f2 > 1024 || n_heads < 1 || ((long long)3 * f2 + (long long)f2 * n_heads + n_heads) * 4 > 64 * 1024)
        return fail(RISVEC_ERR_SHAPE, "%s: f2=%d n_heads=%d (f2 <= 1024, f2*(n_heads+2) floats must fit 64 KB of LDS)", fn,
                    f2, n_heads);
    REQ_PTR(g, "g"); OPT_PTR(b2, "b2"); REQ_PTR(ln_w, "ln_w"); REQ_PTR(ln_b, "ln_b"); REQ_PTR(Wh, "Wh"); REQ_PTR(bh, "bh");
    REQ_PTR(heads, "heads");
    return finish(fn, risvec::launch_policy_heads(n_envs, n_veh, f2, n_heads, g, b2, ln_w, ln_b, Wh, bh, heads,
                                                  (hipStream_t)stream));
}

static int episode_dims(const char* fn, int32_t n_envs, int32_t n_veh) {
    if (n_envs < 1) return fail(RISVEC_ERR_SHAPE, "%s: n_envs=%d must be >= 1", fn, n_envs);
    if (n_veh < 1 || n_veh > RISVEC_MAX_VEH)
        return fail(RISVEC_ERR_SHAPE, "%s: n_veh=%d outside [1,%d]", fn, n_veh, RISVEC_MAX_VEH);
    if ((long long)n_envs * (RISVEC_EP_FIXED + n_veh) >= (1LL << 31))
        return fail(RISVEC_ERR_SHAPE, "%s: n_envs*(%d+n_veh) must stay below 2^31", fn, RISVEC_EP_FIXED);
    return RISVEC_OK;
}

int risvec_episode_clear(int32_t n_envs, int32_t n_veh, double* acc, risvec_stream_t stream) {
    const char* fn = "risvec_episode_clear";
    if (int rc = episode_dims(fn, n_envs, n_veh)) return rc;
    REQ_PTR(acc, "acc");
    return finish(fn, risvec::launch_episode_clear(n_envs, n_veh, acc, (hipStream_t)stream));
}

int risvec_episode_accumulate(int32_t n_envs, int32_t n_veh, const float* metrics, const float* reward,
                              const float* power_w, float user_clip, double* acc, risvec_stream_t stream) {
    const char* fn = "risvec_episode_accumulate";
    if (int rc = episode_dims(fn, n_envs, n_veh)) return rc;
    REQ_PTR(metrics, "metrics"); REQ_PTR(reward, "reward"); OPT_PTR(power_w, "power_w"); REQ_PTR(acc, "acc");
    if (!(user_clip >= 0.0f)) return fail(RISVEC_ERR_ARG, "%s: user_clip=%g must be >= 0", fn, (double)user_clip);
    return finish(fn, risvec::launch_episode_accumulate(n_envs, n_veh, metrics, reward, power_w, user_clip, acc,
                                                        (hipStream_t)stream));
}

int32_t risvec_episode_partial_rows(int32_t n_envs) { return n_envs < 1 ? 0 : risvec::episode_partial_rows(n_envs); }

int risvec_episode_summary(int32_t n_envs, int32_t n_veh, int32_t n_steps, const double* acc, const float* metrics,
                           double* per_env, double* partial, double* summary, risvec_stream_t stream) {
    const char* fn = "risvec_episode_summary";
    if (int rc = episode_dims(fn, n_envs, n_veh)) return rc;
    if (n_steps < 1) return fail(RISVEC_ERR_ARG, "%s: n_steps=%d must be >= 1 (no step was accumulated)", fn, n_steps);
    REQ_PTR(acc, "acc"); REQ_PTR(metrics, "metrics"); OPT_PTR(per_env, "per_env"); REQ_PTR(partial, "partial");
    REQ_PTR(summary, "summary");
    return finish(fn, risvec::launch_episode_summary(n_envs, n_veh, n_steps, acc, metrics, per_env, partial, summary,
                                                     (hipStream_t)stream));
}

}  // extern "C"
