/* risvec.h -- C ABI of the MI355X-native vectorised RIS-VEC environment.
 *
 * This is the drop-in boundary for ONE hot path of 20242204033/RIS-VEC-MARL:
 * `Simulation-MARL-BCD/Environment.py` (ENV below) -- class `Environ`, its reset /
 * mobility / geometry / channel-gain / BCD / step() methods.  The reference has no
 * FFI of its own (the boundary there is a Python class, ENV:56); every entry point
 * below names the reference method it replaces, batched over `n_envs` independent
 * environments.  The Python host side (`ris_vec_marl_amd/`) binds these with ctypes;
 * `INTEGRATION.md` shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - Plain C: device pointers, sizes, POD structs.  No torch / HIP types: a stream
 *     is passed as `void*` (a `hipStream_t`; NULL = the default stream).
 *   - All launches are asynchronous on `stream`.  Return value: RISVEC_OK or an
 *     error code; `risvec_last_error()` gives the message (thread-local).
 *   - Arrays are row-major, E = n_envs, V = n_veh, M = n_ris.  Complex arrays are
 *     interleaved (re, im) float pairs.  Every pointer must be 16-byte aligned.
 *   - Random draws: each entry point takes optional "injected draw" arrays (used by
 *     the parity tests, which replay the reference's own MT19937 draws); when NULL
 *     the kernel draws from Philox4x32-10 keyed by (seed; global env id, vehicle,
 *     counter, site), so results do not depend on how envs are sharded over GPUs.
 */
#ifndef RISVEC_H
#define RISVEC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RISVEC_ABI_VERSION 15
#define RISVEC_POISSON_TABLE 64   /* entries of the arrival CDF table            */
#define RISVEC_MAX_LANES 8        /* lane coordinates per direction (ref. has 4) */
#define RISVEC_MAX_VEH 64         /* V <= 64: one env's vehicles fit a wavefront */
#define RISVEC_METRICS 16         /* floats per env in `metrics` (14 used)       */
#define RISVEC_PARTNER_SINGLE (-1)      /* alone in a 1-element NOMA group (OMA)     */
#define RISVEC_PARTNER_NONE (-2)        /* in no group / a group of another size     */
#define RISVEC_PARTNER_SECOND (1 << 16) /* added to the index when listed 2nd in pair */

enum {
    RISVEC_OK = 0,
    RISVEC_ERR_ARG = 1,         /* NULL / misaligned pointer, bad struct version    */
    RISVEC_ERR_SHAPE = 2,       /* dimension outside what the kernels support       */
    RISVEC_ERR_LAUNCH = 3,      /* HIP reported a launch error                      */
    RISVEC_ERR_UNSUPPORTED = 4
};

enum { RISVEC_DIR_U = 0, RISVEC_DIR_D = 1, RISVEC_DIR_L = 2, RISVEC_DIR_R = 3 };

/* ENV:70, 263, 311-317 */
enum { RISVEC_CH_FREE = 0, RISVEC_CH_3GPP_UMI = 1, RISVEC_CH_3GPP_UMA = 2, RISVEC_CH_OTHER = 3 };

/* step flags */
enum {
    RISVEC_STEP_METRICS = 1,        /* write metrics[E,16] (the 13 last_* + global_reward) */
    RISVEC_STEP_POWER_W = 2,        /* write power_w[E,2,V]  (last_power_W, ENV:664-666)   */
    RISVEC_STEP_POLICY_ACTION = 4,  /* `action` is the policy output [E,V,2] in [-1,1]; apply
                                       marl_train_bcd.py:1601-1608 in-kernel             */
    RISVEC_STEP_OBS = 8,            /* write obs[E,V,5] (marl_train_bcd.py:819-827)        */
    RISVEC_STEP_REUSE_COLSUM = 16,  /* risvec_step_fused_bcd: c_col is current, skip its rebuild */
    RISVEC_STEP_REUSE_SSUM = 32,    /* risvec_step_fused_bcd: s_sum is current (see RISVEC_BCD_REUSE_SSUM) */
    RISVEC_STEP_REUSE_IDX = 128,    /* risvec_step_fused_bcd: theta_idx is current (see RISVEC_BCD_REUSE_IDX) */
    RISVEC_STEP_THETA_BY_INDEX = 256, /* risvec_step_fused_bcd (with RISVEC_STEP_REUSE_IDX, control_bit = 3, a shape for
                                       which risvec_theta_by_index_supported() is 1): theta is kept BY INDEX -- the sweep
                                       updates state.theta_idx and does not write the complex64 state.theta, the fused
                                       step expands the indices through a 9-entry table.  Same outputs bit for bit;
                                       state.theta is stale until risvec_theta_from_index() materialises it. */
    RISVEC_STEP_STEER = 64          /* risvec_step_fused: h_r is the steering vector risvec_geometry wrote
                                       (phases_R_i[v,m] = z_v^m, z_v = exp(-j pi angle_v), ENV:249-253): do not
                                       read it; evaluate sum_m theta_m b_m z^m by Horner in float64 from
                                       state.z_r (16 bytes per vehicle instead of 8M).  Only valid while h_r is
                                       what the geometry kernel produced -- not for arbitrary channel draws. */
};

/* risvec_bcd flags */
enum {
    RISVEC_BCD_REUSE_COLSUM = 1,   /* caller guarantees c_col matches h_r and b */
    RISVEC_BCD_REUSE_SSUM = 2,     /* caller guarantees theta and c_col are unchanged since the last sweep
                                      wrote s_sum: start from it instead of re-summing theta.c */
    RISVEC_BCD_REUSE_IDX = 4,      /* caller guarantees theta is unchanged since the last sweep wrote theta_idx (the
                                      candidate index of every element): control_bit = 3 then takes the indexed sweep,
                                      which never reads theta (two lanes per env, about a third of the instructions per
                                      coordinate) */
    RISVEC_BCD_NO_THETA = 8        /* with RISVEC_BCD_REUSE_IDX and control_bit = 3: update theta_idx only, leave the
                                      complex64 theta to risvec_theta_from_index() */
};

/* Physics / geometry parameters: the attributes of `Environ` that the driver sets
 * (ENV:57-190, overridden by marl_train_bcd.py:548-779).  Passed by value to kernels. */
typedef struct RisVecParams {
    uint32_t abi_version;       /* = RISVEC_ABI_VERSION                               */
    uint32_t struct_bytes;      /* = sizeof(RisVecParams)                             */
    /* PHY  (ENV:72-77, 125, 555) */
    float bandwidth_mhz;
    float noise_power;
    float p_max;
    float power_scale;
    /* QoS  (ENV:79-82) */
    int32_t qos_enable;
    float r_min_bpshz;
    float d_max_s;
    float qos_penalty;
    /* timing / compute  (ENV:101-116) */
    float time_fast;
    float k_cpu;
    float f_local_max;
    float f_edge_max;
    float cycles_per_bit;
    float cpu_share_floor;      /* raw attribute; the kernels apply ENV:574-577       */
    /* reward  (ENV:138-143, 696-703) */
    float w_d;
    float w_e;
    float reward_clip;
    /* arrivals  (ENV:156, 717-719): rate + its CDF table, built on the host in f64    */
    float arrival_rate;
    float poisson_cdf[RISVEC_POISSON_TABLE];
    /* 3GPP modes  (ENV:186-189, 96) */
    float fc_ghz;
    float shadow_std_los;
    float shadow_std_nlos;
    float rician_k_db;
    float veh_ant_gain;
    int32_t n_lanes;            /* entries used in each lane array (reference: 4)     */
    /* geometry, double: positions are advanced in f64 exactly like the reference     */
    double time_slow;           /* ENV:101 */
    double width, height;       /* ENV:63-64 */
    double lanes_up[RISVEC_MAX_LANES];
    double lanes_down[RISVEC_MAX_LANES];
    double lanes_left[RISVEC_MAX_LANES];
    double lanes_right[RISVEC_MAX_LANES];
} RisVecParams;

/* Device-resident state and outputs of E environments (struct-of-arrays).
 * The struct itself lives in HOST memory; the pointers are device pointers. */
typedef struct RisVecState {
    uint32_t abi_version;
    uint32_t struct_bytes;
    int32_t n_envs, n_veh, n_ris, control_bit;
    int64_t env_offset;     /* global id of local env 0 (multi-GPU sharding; RNG key) */
    /* vehicles (ENV:45-53) */
    double *pos;            /* [E,V,2] f64  x,y                                       */
    int32_t *dir;           /* [E,V]   RISVEC_DIR_*                                   */
    float *vel;             /* [E,V]   m/s (integers)                                 */
    /* geometry (ENV:241-253) */
    float *dist_r;          /* [E,V]   distances_R_i                                  */
    float *ang_r;           /* [E,V]   angles_R_i                                     */
    float *pl;              /* [E,V]   ro^2 / (d_Rv^2.2 d_BR^2.5)   (ENV:270-272)     */
    float *h_r;             /* [E,V,M] c64  phases_R_i                                */
    /* RIS (ENV:171-179) */
    float *theta;           /* [E,M]   c64  elements_phase_shift_complex              */
    const float *b;         /* [M]     c64  phase_R (shared by all envs)              */
    const float *h_d;       /* [E,V]   c64  optional direct link (NULL = 0, as ref.)  */
    float *gain;            /* [E,V]   channel_gains                                  */
    /* queues (ENV:116, 151) */
    float *data_buf;        /* [E,V]   DataBuf, kbit                                  */
    float *mec_q;           /* [E]     mec_queue_cycles                               */
    /* step outputs (ENV:731 + attributes read by the driver) */
    float *rate;            /* [E,V]   vehicle_rate                                   */
    float *data_t;          /* [E,V]                                                  */
    float *data_p;          /* [E,V]                                                  */
    float *reward;          /* [E,V]   per_user_reward                                */
    float *over_power;      /* [E,V]                                                  */
    float *obs;             /* [E,V,5] marl_get_state                                 */
    float *metrics;         /* [E,16]  see RISVEC_METRIC_* (slots 14,15 reserved = 0) */
    float *power_w;         /* [E,2,V] last_power_W (may be NULL unless flag set)     */
    /* BCD cache: c[e,m] = (sum_v h_r[e,v,m]) * b[m] in float64 (pure geometry, like `pl`) */
    double *c_col;          /* c128, ceil(E/64)*64*M elements, LANE-MAJOR: (e,m) at ((e/64)*M+m)*64+e%64;
                               written by risvec_geometry / risvec_colsum, read only by risvec_bcd   */
    double *s_sum;          /* [E]     c128; S = sum_m theta_m c_m left by the last BCD sweep (may be NULL) */
    /* SARL variant only (Simulation-SARL/Environment.py:337-340); the MARL step never writes it */
    float *over_data;       /* [E,V]   (may be NULL for MARL-only use)                 */
    /* steering base z[e,v] = exp(-j pi angle_R[e,v]) in float64 (h_r[e,v,m] = z^m); written by risvec_geometry
       when non-NULL, read by risvec_step_fused with RISVEC_STEP_STEER */
    double *z_r;            /* [E,V]   c128 (may be NULL)                                 */
    /* Candidate index of every theta element as the last sweep left it (0..7 = exp(j 2 pi k / 8), ENV:169, 213;
       8 = the integer 0 of ENV:211, 220), one byte each, row-major [E][32*ceil(M/32)] (rows padded to a multiple of
       32 bytes); written by every control_bit = 3 sweep, read with RISVEC_BCD_REUSE_IDX / RISVEC_STEP_THETA_BY_INDEX
       (may be NULL) */
    uint8_t *theta_idx;
} RisVecState;

/* Parameters of the single-agent (SARL) environment variant,
 * Simulation-SARL/Environment.py:66-83 (SENV). */
typedef struct RisVecSarlParams {
    uint32_t abi_version;
    uint32_t struct_bytes;
    float time_fast;        /* SENV:67 */
    float bandwidth_mhz;    /* SENV:69 */
    float k_cpu;            /* SENV:70 */
    float cycles_l;         /* SENV:71  L */
    float t_factor1;        /* SENV:80 */
    float t_factor2;        /* SENV:81 */
    float penalty1;         /* SENV:82 */
    float penalty2;         /* SENV:83 */
    float arrival_rate;     /* SENV:78 */
    float poisson_cdf[RISVEC_POISSON_TABLE];
} RisVecSarlParams;

/* metrics slots (ENV line in comment) */
enum {
    RISVEC_METRIC_GLOBAL_REWARD = 0,   /* 721 */
    RISVEC_METRIC_OFF_KBIT_SUM = 1,    /* 612 */
    RISVEC_METRIC_LOCAL_KBIT_SUM = 2,  /* 613 */
    RISVEC_METRIC_MEC_QUEUE = 3,       /* 614 */
    RISVEC_METRIC_BACKLOG_MEAN = 4,    /* 649 */
    RISVEC_METRIC_DELAY_LOCAL = 5,     /* 643 */
    RISVEC_METRIC_DELAY_EDGE_Q = 6,    /* 644 */
    RISVEC_METRIC_DELAY_EDGE_C = 7,    /* 645 */
    RISVEC_METRIC_T_TX = 8,            /* 646 */
    RISVEC_METRIC_MEC_UTIL = 9,        /* 653 */
    RISVEC_METRIC_LOCAL_UTIL = 10,     /* 656 */
    RISVEC_METRIC_QOS_VIOL = 11,       /* 677 */
    RISVEC_METRIC_DELAY = 12,          /* 710 */
    RISVEC_METRIC_ENERGY = 13          /* 711 */
};

typedef void *risvec_stream_t;

uint32_t risvec_abi_version(void);
/* Name of the kernel the calling thread's last risvec_step* / risvec_bcd call dispatched ("" before the first): which
 * member of the fused-step family a shape / batch size takes is a dispatch decision; tests assert it. */
const char *risvec_last_kernel(void);
const char *risvec_last_error(void);

/* Fill `p` with the class defaults of ENV:57-190 and the reference driver's lanes
 * (marl_train_bcd.py:446-449).  Host-only helper. */
void risvec_default_params(RisVecParams *p);

/* make_new_game (ENV:733-737) + add_new_vehicles_by_number (ENV:381-410).
 * spawn_ints [E,V,3] int32 = (aux, coord, velocity), buf0 [E] int32 (the single
 * randint(5,9) of ENV:737); both NULL -> Philox.  Writes pos, dir, vel, data_buf.
 * Like the reference it does NOT touch mec_q, theta, gain. */
int risvec_reset(const RisVecState *s, const RisVecParams *p, const int32_t *spawn_ints,
                 const int32_t *buf0, uint64_t seed, uint32_t counter, risvec_stream_t stream);

/* renew_positions (ENV:412-542).  u_turn [E,V,8] float32 uniform draws consumed left
 * to right, one per detected lane crossing; NULL -> Philox.  n_used [E,V] int32
 * (optional) receives the number of draws consumed. */
int risvec_mobility(const RisVecState *s, const RisVecParams *p, const float *u_turn,
                    int32_t *n_used, uint64_t seed, uint32_t counter, risvec_stream_t stream);

/* compute_parms (ENV:241-253): pos -> dist_r, ang_r, pl, h_r. */
int risvec_geometry(const RisVecState *s, const RisVecParams *p, risvec_stream_t stream);

/* update_channel_gains, "free" model (ENV:263-273): theta, h_r, b, pl -> gain. */
int risvec_gain(const RisVecState *s, const RisVecParams *p, risvec_stream_t stream);

/* update_channel_gains, 3GPP modes (ENV:275-327).  model = RISVEC_CH_*.  Injected
 * draws (all [E,V] float32, all or none): u_los ~ U[0,1), z_shadow ~ N(0,1),
 * small = small-scale power (ENV:13-25).  NULL -> Philox (Rayleigh or Rice per
 * p->rician_k_db). */
int risvec_gain_3gpp(const RisVecState *s, const RisVecParams *p, int32_t model,
                     const float *u_los, const float *z_shadow, const float *small,
                     uint64_t seed, uint32_t counter, risvec_stream_t stream);

/* Column sums for BCD: c_col[e,m] = (sum_v h_r[e,v,m]) * b[m], float64.  One HBM pass over
 * h_r.  risvec_geometry calls it too, so after compute_parms the cache is current; call it
 * again after writing h_r directly. */
int risvec_colsum(const RisVecState *s, risvec_stream_t stream);

/* optimize_phase_shift (ENV:208-220) with the objective of ENV:222-231: one BCD sweep
 * over theta in place.  Rebuilds c_col first unless flags has RISVEC_BCD_REUSE_COLSUM.
 * idx_out [E,M] int32 (optional): chosen candidate, -1 = none. */
int risvec_bcd(const RisVecState *s, const RisVecParams *p, int32_t *idx_out, uint32_t flags,
               risvec_stream_t stream);

/* get_next_phase (ENV:233-239): theta = exp(j*angle), angle [E,M] float32. */
/* state.theta[e,m] = the complex64 image of candidate state.theta_idx[e,m]: materialises theta after sweeps that kept
 * it by index (RISVEC_BCD_NO_THETA / RISVEC_STEP_THETA_BY_INDEX).  What those sweeps would have stored, bit for bit. */
int risvec_theta_from_index(const RisVecState *s, risvec_stream_t stream);
/* 1 when the fused step has a theta-by-index form for this shape (RISVEC_STEP_THETA_BY_INDEX), else 0 */
int risvec_theta_by_index_supported(int32_t n_veh, int32_t n_ris);

int risvec_set_phase(const RisVecState *s, const float *angle, risvec_stream_t stream);

/* Random_phase (ENV:203-206): theta = exp(j*possible_angles[idx]); idx [E,M] int32 or
 * NULL -> Philox. */
int risvec_random_phase(const RisVecState *s, const int32_t *idx, uint64_t seed,
                        uint32_t counter, risvec_stream_t stream);

/* step (ENV:547-731) with compute_data_rate (ENV:331-372), using the cached `gain`.
 * action [E,2,V] float32 (or [E,V,2] policy output with RISVEC_STEP_POLICY_ACTION),
 * partner [E,V] int32 + n_groups [E] int32 encode `noma_groups`, arrivals [E,V] int32
 * = the Poisson draws of ENV:718 (NULL -> Philox, counter = step index). */
int risvec_step(const RisVecState *s, const RisVecParams *p, const float *action,
                const int32_t *partner, const int32_t *n_groups, const int32_t *arrivals,
                uint64_t seed, uint32_t counter, uint32_t flags, risvec_stream_t stream);

/* compute_data_rate (ENV:331-372) on its own: p_off [E,V] float32 = offload power in W
 * (row 0 of the reference's `power`), cached `gain` -> rate_out [E,V] bit/s/Hz. */
int risvec_data_rate(const RisVecState *s, const RisVecParams *p, const float *p_off,
                     const int32_t *partner, const int32_t *n_groups, float *rate_out,
                     risvec_stream_t stream);

/* The fused north-star kernel: update_channel_gains ("free") + step in ONE launch:
 * one HBM pass over h_r and theta, gains exchanged in LDS, also written to `gain`. */
int risvec_step_fused(const RisVecState *s, const RisVecParams *p, const float *action,
                      const int32_t *partner, const int32_t *n_groups, const int32_t *arrivals,
                      uint64_t seed, uint32_t counter, uint32_t flags, risvec_stream_t stream);

/* Trajectory record of a multi-step launch: what the driver reads after every step (per-user rewards
 * TRAIN:1611, the observation TRAIN:819-827, the metrics row TRAIN:1627-1662), one slice per step.
 * Device pointers; any of them may be NULL (not recorded). */
typedef struct RisVecTraj {
    float *reward;          /* [T,E,V]                                                 */
    float *obs;             /* [T,E,V,5]                                               */
    float *metrics;         /* [T,E,16]                                                */
} RisVecTraj;

/* The T-step launch: exactly `n_steps` consecutive risvec_step_fused calls -- the driver's step loop
 * marl_train_bcd.py:1304-1611 between two channel refreshes, with the NOMA groups frozen as they are inside an
 * episode -- in ONE launch.  actions [T,E,2,V] (or [T,E,V,2] with RISVEC_STEP_POLICY_ACTION); arrivals [T,E,V]
 * int32 or NULL (Philox, counter + t at step t, i.e. the counters T single calls would use); partner /
 * n_groups are the same for every step.  The state tensors end up exactly as after the last of the T single calls
 * (bit for bit); every step's reward / obs / metrics additionally go to traj (may be NULL).  h_r and theta
 * cannot change inside the launch, so the cascaded gains are computed once and each env's queues stay in
 * registers: at small batches (BASELINE configs[1]) this removes the per-step launch and memory round trip
 * that bound a single-step launch.  RISVEC_STEP_REUSE_* / RISVEC_STEP_STEER are not accepted here. */
int risvec_step_fused_multi(const RisVecState *s, const RisVecParams *p, int32_t n_steps, const float *actions,
                            const int32_t *partner, const int32_t *n_groups, const int32_t *arrivals,
                            uint64_t seed, uint32_t counter, uint32_t flags, const RisVecTraj *traj,
                            risvec_stream_t stream);

/* The same on the CACHED gains (state.gain as risvec_gain / a fused step left it): exactly `n_steps` consecutive
 * risvec_step calls in one launch, for any shape -- the reference driver's own cadence, step() every step and the
 * channel gains only every K_STEPS_FOR_RIS_OPTIMIZATION = 100 steps (marl_train_bcd.py:1304-1611, 1307-1309).
 * Arguments as risvec_step_fused_multi; h_r / theta are not read. */
int risvec_step_multi(const RisVecState *s, const RisVecParams *p, int32_t n_steps, const float *actions,
                      const int32_t *partner, const int32_t *n_groups, const int32_t *arrivals,
                      uint64_t seed, uint32_t counter, uint32_t flags, const RisVecTraj *traj,
                      risvec_stream_t stream);

/* SARL variant: Simulation-SARL/Environment.py step(action_power, action_phase)
 * SENV:321-359 for every env: get_next_phase (theta = exp(j*action_phase), action_phase [E,M]
 * float32 radians; NULL keeps the current theta), the RIS cascaded gain, natural-log rate against
 * sigma^2 = 1e-14, cube-root local-CPU model, buffer-length reward with its two penalties,
 * Poisson arrivals.  action_power [E,2,V] is used as given (no projection).  Outputs: data_buf,
 * data_t, data_p, over_power, over_data, rate, reward [E,V] and the mean reward in
 * metrics[e,0]; with RISVEC_STEP_OBS also obs[E,V,5] = the non-theta tail of ddpg_train.py:47-73. */
int risvec_sarl_step(const RisVecState *s, const RisVecSarlParams *p, const float *action_power,
                     const float *action_phase, const int32_t *arrivals, uint64_t seed,
                     uint32_t counter, uint32_t flags, risvec_stream_t stream);

/* BCD sweep + gains + step in one launch (BASELINE config 5: h_r read once into LDS). */
int risvec_step_fused_bcd(const RisVecState *s, const RisVecParams *p, const float *action,
                          const int32_t *partner, const int32_t *n_groups,
                          const int32_t *arrivals, uint64_t seed, uint32_t counter,
                          uint32_t flags, risvec_stream_t stream);

/* ===========================================================================================
 * NOMA grouping stage: the pairing the reference driver computes right before
 * every env.step() -- marl_train_bcd.py (TRAIN): feasibility mask TRAIN:128-156 + 842-855, score
 * matrix TRAIN:164-194, quantile-gated max-weight matching TRAIN:326-398, greedy completion
 * TRAIN:276-324, mask relaxation TRAIN:260-275, pair QoS check TRAIN:858-880, and the per-step
 * control logic with the freeze-in-episode safeties TRAIN:1401-1562, 1618-1623 -- for all E envs,
 * one wavefront per env, every comparison in float64 in the reference's association order.
 * n_veh <= 16.  Its outputs (partner / n_groups) are exactly what risvec_step* consume.
 * =========================================================================================== */
#define RISVEC_NOMA_MAX_VEH 16

typedef struct RisVecNomaParams {
    int32_t min_pair_target;        /* TRAIN:489   max(1, n_veh / 4); config.yaml 3        */
    int32_t mwm_backoff_rounds;     /* TRAIN:440 / 659                                      */
    int32_t mwm_allow_singles;      /* TRAIN:436                                            */
    int32_t qos_enable;             /* TRAIN:750   soft QoS penalty in the score            */
    int32_t relax_topk_step;        /* TRAIN:728                                            */
    int32_t freeze_group_in_episode;/* TRAIN:738                                            */
    int32_t freeze_recalc_every;    /* TRAIN:739   0 = frozen for the whole episode         */
    int32_t mask_enable;            /* TRAIN:498   0: pairing always sees the full mask     */
    double mwm_accept_quantile;     /* TRAIN:439                                            */
    double mwm_accept_q_step;       /* TRAIN:441                                            */
    double completion_min_quantile; /* TRAIN:282, 300                                       */
    double score_w_delta_db;        /* TRAIN:716                                            */
    double abs_gain_min_db;         /* TRAIN:723   (-inf = off)                             */
    double qos_soft_penalty;        /* TRAIN:172, 1450                                      */
    double qos_R_min;               /* TRAIN:751   bit/s/Hz                                 */
    double noise_power;             /* env.noise_power (ENV:72-76)                          */
    double P_max;                   /* env.P_max (ENV:125)                                  */
    double relax_tau_factor;        /* TRAIN:729                                            */
    double tau_back_floor_db;       /* TRAIN:1499                                           */
    double freeze_reward_drop_ratio;/* TRAIN:741                                            */
    double freeze_unstick_prob;     /* TRAIN:740                                            */
    float score_w_history;          /* TRAIN:717   (float32 product with the float32 history) */
    float pair_hist_decay;          /* TRAIN:719   (float32 in-place decay)                 */
} RisVecNomaParams;

enum RisVecNomaFlag {               /* bits of RisVecNomaState.flags[e]                     */
    RISVEC_NOMA_HAS_LAST = 1,       /* last_env_global is set (TRAIN:1298, 1618)            */
    RISVEC_NOMA_UNSTICK_USED = 2,   /* unstick_used_flag (TRAIN:1300, 1536)                 */
    RISVEC_NOMA_HAS_GROUPS = 4      /* episode_groups is set (TRAIN:1297, 1552)             */
};

/* Episode-scoped state of TRAIN:1282-1300, device pointers, N = n_veh.  Most steps of an episode
 * are frozen steps (TRAIN:1542-1547): for those risvec_noma_group only counts the step in `pending`;
 * the history decay / pair increments (TRAIN:1406, 1556-1558) and streak updates (TRAIN:1560-1561)
 * they owe are replayed, operation for operation, when the env next re-solves its pairing or when
 * risvec_noma_flush is called.  hist / streak are therefore current only after a flush; partner /
 * n_groups / flags always are. */
typedef struct RisVecNomaState {
    int32_t n_envs, n_veh;
    int64_t env_offset;             /* global id of local env 0 (RNG key)                   */
    float *hist;                    /* [E,N,N] pair_affinity_hist                           */
    int32_t *streak;                /* [E,N]   unpaired_streak                              */
    int32_t *partner;               /* [E,N]   episode_groups = this step's noma_groups, partner encoding of risvec_step */
    int32_t *n_groups;              /* [E]     len(episode_groups)                          */
    double *last_global;            /* [E]     last_env_global                              */
    double *best_global;            /* [E]     ep_env_best                                  */
    uint8_t *flags;                 /* [E]     RisVecNomaFlag bits                          */
    uint8_t *mask;                  /* [E,N,N] last_mask_mat (0/1)                          */
    double *tau;                    /* [E]     last_tau_now                                 */
    int32_t *pending;               /* [E]     frozen steps not yet applied to hist / streak */
    void *scratch;                  /* risvec_noma_scratch_bytes(E, N) bytes of ZEROED device memory (NULL when that is 0):
                                       beyond 8 vehicles risvec_noma_group is two launches, and this is the list of envs
                                       the first leaves to the second (a pairing graph too dense for its on-chip table) */
    int64_t scratch_bytes;
} RisVecNomaState;

/* Scratch risvec_noma_group needs for this batch: 0 up to 8 vehicles, 16 B + 4 B per env (rounded up to 256 B) beyond.
 * Zero it once when it is allocated (risvec_noma_begin_episode also empties the list); every call leaves it empty. */
int64_t risvec_noma_scratch_bytes(int32_t n_envs, int32_t n_veh);

void risvec_noma_default_params(RisVecNomaParams *p, int32_t n_veh);   /* driver Config defaults */

/* Start of an episode (TRAIN:1282-1300): zero hist / streak / flags / pending. */
int risvec_noma_begin_episode(const RisVecNomaState *ns, risvec_stream_t stream);

/* Channel-refresh step (TRAIN:1319-1343): tau = quantile q_now of |g_strong - g_weak| in dB
 * (TRAIN:842-855) -> ns->tau; with K_now >= 1 also the N x N mask (TRAIN:134-156) -> ns->mask.
 * gain [E,N] float32 linear; gdb15 [E,N] float64 = 10 log10(max(g, 1e-15)) or NULL (computed on
 * the device; pass it to reproduce a host's log10 bit for bit -- see EXPERIMENTS.md, section 8 f2). */
int risvec_noma_mask(const RisVecNomaState *ns, const float *gain, const double *gdb15, double q_now,
                     int32_t K_now, risvec_stream_t stream);

/* One pass of TRAIN:1401-1562 for every env; this step's noma_groups are left in ns->partner /
 * ns->n_groups, ready for risvec_step*.
 *   gain [E,N] f32; gdb12 [E,N] f64 = 10 log10(max(g, 1e-12)) or NULL; p_off01 [E,N] f32 = the
 *   offload power in [0,1] used by the QoS check (TRAIN:1391-1396; may be NULL when qos is off);
 *   use_mask: this step's mask_mat is ns->mask (a refresh step) / 0 = None (TRAIN:1421-1424);
 *   K_back, tau_back [E]: last_K_now / last_tau_now (TRAIN:1486-1491; last_q_now only feeds a
 *   value the reference computes and never uses, TRAIN:1497);
 *   prev_global (stride in floats) or NULL: global reward of the PREVIOUS step -- the
 *   ep_env_best / last_env_global bookkeeping of TRAIN:1618-1623 is applied first;
 *   u_unstick [E] f32 or NULL (Philox) : the draw of TRAIN:1539;
 *   info_out [E,4] or NULL: {recomputed, back-off rounds, pairs, matchable users of the last matching}. */
int risvec_noma_group(const RisVecNomaState *ns, const RisVecNomaParams *np, const float *gain,
                      const double *gdb12, const float *p_off01, int32_t use_mask, int32_t K_back,
                      const double *tau_back, const float *prev_global, int32_t prev_global_stride,
                      int32_t i_step, const float *u_unstick, uint64_t seed, uint32_t counter,
                      int32_t *info_out, risvec_stream_t stream);

/* risvec_noma_group reading the pairing power straight from the SAC power head: power_raw [E,N,2] in [-1,1], of which
 * component 0 is mapped as risvec_marshal_actions maps it (TRAIN:1391-1396, bit for bit) -- no marshalling launch. */
int risvec_noma_group_raw(const RisVecNomaState *ns, const RisVecNomaParams *np, const float *gain,
                          const double *gdb12, const float *power_raw, int32_t use_mask, int32_t K_back,
                          const double *tau_back, const float *prev_global, int32_t prev_global_stride,
                          int32_t i_step, const float *u_unstick, uint64_t seed, uint32_t counter,
                          int32_t *info_out, risvec_stream_t stream);

/* Apply the deferred frozen steps to hist / streak (pair_hist_decay = np->pair_hist_decay). */
int risvec_noma_flush(const RisVecNomaState *ns, const RisVecNomaParams *np, risvec_stream_t stream);

/* ===========================================================================================
 * Replay ring buffer + policy-output marshalling: buffer.py (BUF) kept in HBM and
 * fed straight from the step kernel's outputs, and the marshalling the driver does around
 * env.step() (TRAIN:1386-1396, 1601-1608, 1776-1799).
 * =========================================================================================== */
typedef struct RisVecReplay {            /* BUF:4-14; device pointers, row r of every array = transition r */
    int32_t n_agents, input_shape, n_actions;   /* state row = input_shape*n_agents floats, action row = n_actions*n_agents */
    int32_t reserved;
    int64_t mem_size;
    float *state_memory;                /* [mem_size, input_shape*n_agents]  */
    float *action_memory;               /* [mem_size, n_actions*n_agents]    */
    float *reward_global_memory;        /* [mem_size]                        */
    float *reward_local_memory;         /* [mem_size, n_agents]              */
    float *new_state_memory;            /* [mem_size, input_shape*n_agents]  */
    uint8_t *terminal_memory;           /* [mem_size] 0/1                    */
    float *mask_memory;                 /* [mem_size, n_agents*n_agents]     */
} RisVecReplay;

/* n consecutive store_transition calls (BUF:16-25), transition e landing in row (mem_cntr + e) % mem_size;
 * the caller advances its mem_cntr by n.  n <= mem_size.  reward_g is read with a stride (in floats) so
 * metrics[:,0] can be passed as is; done [n] 0/1 or NULL (then done_all applies to every transition);
 * mask [n, A*A] 0/1 bytes (the NOMA mask) or NULL = all ones (TRAIN:1786-1787); state_carry (or NULL)
 * receives a copy of state_ -- the next step's `state` (marl_state_old_all, TRAIN:1277) without a
 * separate copy; it must not alias state / state_. */
int risvec_replay_store(const RisVecReplay *rb, int64_t mem_cntr, int32_t n, const float *state, const float *action,
                        const float *reward_g, int32_t reward_g_stride, const float *reward_l, const float *state_,
                        const uint8_t *done, int32_t done_all, const uint8_t *mask, float *state_carry,
                        risvec_stream_t stream);

/* The rollout's transition store FUSED into the step (round 3): one launch runs step() (fused != 0: with the RIS
 * cascaded gains, as risvec_step_fused; 0: on the cached gains, as risvec_step) and appends this step's n_envs
 * transitions to the ring -- what risvec_step* followed by risvec_replay_store_policy do in two launches, bit for bit
 * (marl_train_bcd.py:1776-1799, buffer.py:16-25):
 *   state      <- state.obs as the kernel finds it (the observation the policy acted on: keep it current),
 *   action     <- per agent [probs_i with zero diagonal | raw power_i] from `action` (the raw policy output [E,V,2]:
 *                 flags must hold RISVEC_STEP_POLICY_ACTION | RISVEC_STEP_OBS) and ring->probs [E,V,V],
 *   reward_l / reward_g / state_ <- this step's per-user rewards, their mean, the new observation,
 *   done       <- ring->done for every transition, mask <- ring->mask [E,V,V] bytes or all ones when NULL.
 * Needs n_veh in {4, 8, 16}, rb.n_agents = n_veh, rb.input_shape = 5, rb.n_actions = n_veh + 2, n_envs <= mem_size;
 * fused != 0 additionally needs a shape with a software-pipelined kernel (RISVEC_ERR_UNSUPPORTED otherwise: use
 * the two-launch form).  The caller advances its mem_cntr by n_envs. */
typedef struct RisVecStepRing {
    RisVecReplay rb;
    int64_t mem_cntr;
    const float *probs;
    const uint8_t *mask;
    int32_t done;
    int32_t reserved;
} RisVecStepRing;
int risvec_step_ring(const RisVecState *s, const RisVecParams *p, const RisVecStepRing *ring, const float *action,
                     const int32_t *partner, const int32_t *n_groups, const int32_t *arrivals, uint64_t seed,
                     uint32_t counter, uint32_t flags, int32_t fused, risvec_stream_t stream);

/* risvec_replay_store with the action row built in the store kernel from the policy outputs -- power_raw [n,A,2],
 * probs [n,A,A]: per agent [probs_i with zero diagonal | raw power_i], exactly the action_store row of
 * risvec_marshal_actions (TRAIN:1386-1390, 1776-1784) -- so a rollout step needs no marshalling launch: the env
 * takes power_raw with RISVEC_STEP_POLICY_ACTION, the grouping takes it through risvec_noma_group_raw.
 * Needs n_actions = n_agents + 2. */
int risvec_replay_store_policy(const RisVecReplay *rb, int64_t mem_cntr, int32_t n, const float *state,
                               const float *power_raw, const float *probs, const float *reward_g,
                               int32_t reward_g_stride, const float *reward_l, const float *state_, const uint8_t *done,
                               int32_t done_all, const uint8_t *mask, float *state_carry, risvec_stream_t stream);

/* sample_buffer (BUF:27-37): rows idx[b] (int64, each < max_mem = min(mem_cntr, mem_size)) or, with
 * idx NULL, Philox(seed; b, 0, counter, site 8) -> floor(x * max_mem / 2^32); outputs [batch, ...] in the
 * order sample_buffer returns them; idx_out (or NULL) receives the rows used. */
int risvec_replay_sample(const RisVecReplay *rb, int64_t max_mem, int32_t batch, const int64_t *idx, uint64_t seed,
                         uint32_t counter, float *states, float *actions, float *rewards_g, float *rewards_l,
                         float *states_, uint8_t *dones, float *masks, int64_t *idx_out, risvec_stream_t stream);

/* Policy outputs -> the three places the driver sends them: power_raw [E,V,2] (SAC power head, in
 * [-1,1]), probs [E,V,V] (intent probabilities) ->
 *   action_env   [E,2,V]      TRAIN:1601-1608  clip to +-0.999, (x+1)/2, CPU share floored at clamp(floor,0,0.95)
 *   p_off01      [E,V]        TRAIN:1391-1396  (input of the NOMA QoS check)
 *   action_store [E,V*(V+2)]  TRAIN:1386-1390, 1776-1784  per agent [probs_i with zero diagonal, raw power_i]
 * any output may be NULL; probs may be NULL when action_store is. */
int risvec_marshal_actions(int32_t n_envs, int32_t n_veh, const float *power_raw, const float *probs,
                           float cpu_share_floor, float *action_env, float *p_off01, float *action_store,
                           risvec_stream_t stream);

/* Batched choose_action, sampling epilogue (sac_agent.py:80-131, 187-225) + marshalling in one
 * launch.  heads [V, E, 4+V] float32 = per agent the rows (mu[2], log_std[2], intent_logits[V]) its
 * PolicyNetwork.forward produced (sac_agent.py:62-78: three small GEMMs, library work); mask [E,V,V] 0/1
 * bytes or NULL; tau [V] the agents' Gumbel temperatures; hard [V] bytes or NULL: per agent the straight-through
 * one-hot form of F.gumbel_softmax (marl_train_bcd.py:1816-1818); eps [E,V,2] ~ N(0,1) and expo [E,V,V] ~ Exp(1)
 * inject the draws of Normal.sample / F.gumbel_softmax (NULL: Philox keyed by env_offset + e).
 * Outputs: power_raw [E,V,2] (tanh-squashed), probs [E,V,V] (soft Gumbel-softmax), onehot [E,V,V] or
 * NULL; and, each optional, exactly what risvec_marshal_actions would produce from them:
 * action_env [E,2,V], p_off01 [E,V], action_store [E,V*(V+2)]. */
int risvec_policy_sample(int32_t n_envs, int32_t n_veh, int64_t env_offset, const float *heads, const uint8_t *mask,
                         const float *tau, const uint8_t *hard, const float *eps, const float *expo, uint64_t seed,
                         uint32_t counter, float cpu_share_floor, float *power_raw, float *probs, float *onehot,
                         float *action_env, float *p_off01, float *action_store, risvec_stream_t stream);

/* The two non-GEMM ends of PolicyNetwork.forward (sac_agent.py:62-78), all agents and envs per launch;
 * the fc1 x fc2 product in between is a plain batched GEMM (rocBLAS).  Row-major float32 everywhere.
 *   layer1: obs [E,V,in] , W1 [V,in,F1] (= fc1.weight^T), b1 / ln_w / ln_b [V,F1]
 *           -> out [V,E,F1] = relu(LayerNorm(fc1(obs)))                      (F1 <= 1024; in*F1 must fit LDS)
 *   heads : g [V,E,F2] (= fc2 product; b2 [V,F2] its bias, added here, or NULL), ln_w / ln_b [V,F2], Wh [V,F2,H] (= [mu|log_std|intent_logits]
 *           weights transposed, H = 4 + V), bh [V,H]
 *           -> heads [V,E,H], the input of risvec_policy_sample                (F2 <= 1024) */
int risvec_policy_layer1(int32_t n_envs, int32_t n_veh, int32_t in_dims, int32_t f1, const float *obs, const float *W1,
                         const float *b1, const float *ln_w, const float *ln_b, float *out, risvec_stream_t stream);
/* risvec_policy_layer1 writing the hidden row as the float16 operand of a split-precision GEMM: out16
 * [V, E, 3*f1] halfs, row = [ hi(h) | hi(h) 2^-5 | (h - hi(h)) 2^6 ] with hi = round-to-nearest float16.
 * Multiplied (float16 inputs, float32 accumulation) by the fc2 weight stacked along K as
 * [ hi(W) ; (W - hi(W)) 2^5 ; hi(W) 2^-6 ] it gives h W to 2^-22 relative per product -- the accuracy of the
 * float32 GEMM at the fp16 matrix-core rate.  f1 must be a multiple of 4, in_dims <= 8.  |h| must stay below
 * the float16 maximum 65504 (LayerNorm outputs do). */
int risvec_policy_layer1_split16(int32_t n_envs, int32_t n_veh, int32_t in_dims, int32_t f1, const float *obs,
                                 const float *W1, const float *b1, const float *ln_w, const float *ln_b, void *out16,
                                 risvec_stream_t stream);
/* The whole PolicyNetwork.forward (sac_agent.py:62-78) of every agent in ONE launch, all three layers on the
 * fp16 matrix cores at float32 accuracy (every operand split into float16 hi + lo, the three significant partial
 * products accumulated in float32); neither hidden layer touches HBM.  Prepared weights (rebuild after an update),
 * "fragment" = the 8 halfs one lane feeds to v_mfma_f32_32x32x16_f16, S_0 = hi(2^s X), S_1 = fp16(2^s X - S_0), s a
 * per-agent power of two that keeps S_1 in the float16 normal range:
 *   G    [V, 6, 6]  C C^T / f1, C = fc1 weight rows (and the bias as row in_dims) centred over the feature axis:
 *                   LayerNorm-1 variance of an env in closed form, var = x^T G x with x = (obs, 1, 0..);
 *   fc1 operand     X = [f1, 16]: column k < in_dims = C[k] ln1_w, column in_dims = C[in_dims] ln1_w, column in_dims + 1
 *                   = ln1_b, rest 0 (multiplied by (obs rstd, rstd, 1, 0..) it gives the normalised pre-activation);
 *                   fragments of group g (32 features): element (t, lane, j) = S_t[32g + (lane&31)][8(lane>>5) + j];
 *   W1F  [V, 2, 64, 8]  the fc1-operand fragments of group 0;
 *   W2f  [V, f1/32, 8 + 4 f2/32, 64, 8]  the weight stream, per group g of 32 hidden features: 8 fragment rows holding
 *                   the fc1-operand fragments of group g+1 (rows 0-1; rest padding), then for k-steps u = 0, 1 the
 *                   fc2 weight X = W2 [f1, f2] in the k order in which the fc2 MFMA reads the fc1 MFMA's accumulator:
 *                   rows [t][m], element (lane, j) = S_t[32g + 16u + 8(j>>2) + 4(lane>>5) + (j&3)][32m + (lane&31)];
 *   w_unscale [V]   2^-(s_fc1 + s_fc2);
 *   WhF  [V, f2/32, 2, 2, 64, 8]  X = the head weight Wh [f2, n_heads] (columns mu | log_std | intent_logits) zero-padded
 *                   to 32 columns: element (m, u, t, lane, j) = S_t[32m + 16u + 8(j>>2) + 4(lane>>5) + (j&3)][lane&31];
 *   wh_unscale [V]  its 2^-s.
 * heads [V, E, n_heads].  Built for in_dims <= 5, f1 % 32 == 0 <= 1024, f2 in {128, 256}, n_heads <= 24
 * (risvec_policy_mlp_supported); other shapes return RISVEC_ERR_UNSUPPORTED -- use the three-launch form. */
int risvec_policy_mlp_supported(int32_t in_dims, int32_t f1, int32_t f2, int32_t n_heads);
int risvec_policy_mlp(int32_t n_envs, int32_t n_veh, int32_t in_dims, int32_t f1, int32_t f2, int32_t n_heads, const float *obs,
                      const float *G, const void *W1F, const void *W2f, const float *w_unscale, const float *b2,
                      const float *ln2_w, const float *ln2_b, const void *WhF, const float *wh_unscale, const float *bh,
                      float *heads, risvec_stream_t stream);
int risvec_policy_heads(int32_t n_envs, int32_t n_veh, int32_t f2, int32_t n_heads, const float *g, const float *b2,
                        const float *ln_w, const float *ln_b, const float *Wh, const float *bh, float *heads,
                        risvec_stream_t stream);

/* ---- f4: the per-episode metrics sink --------------------------------------------------
 * The driver sums the `last_*` scalars, the clipped per-user rewards and the equivalent powers step by
 * step in Python floats and turns them into per-episode scalars for TensorBoard (marl_train_bcd.py,
 * TRAIN below: 1611-1662 accumulate, 1714 clip, 1717-1753 power, 1769 per-user sums, 1824 and 1838-1865
 * means, 1939-1941 min / var / Jain, 1927-2048 the tags).  Here every env keeps its own float64
 * accumulators on the device and the episode summary is reduced over the envs in two launches.
 *
 * acc [RISVEC_EP_FIXED + V, E] doubles (column-major: one row per quantity, envs contiguous), rows:
 *   0..13   sum over the episode's steps of metrics[slot]            (TRAIN:1626-1662)
 *   14      sum of sum_v power_w[0,v]  (offload, equivalent watts)    (TRAIN:1752)
 *   15      sum of sum_v power_w[1,v]  (local)                        (TRAIN:1753)
 *   16      max over the steps of the global reward (`ep_env_best`)   (TRAIN:1613-1622)
 *   17+v    sum of clip(reward_v, -user_clip, +user_clip)             (TRAIN:1714, 1769)            */
#define RISVEC_EP_FIXED 17
#define RISVEC_EP_COLS 21
enum {
    RISVEC_EP_GLOBAL_AVG = 0,      /* reward/global_avg            mean of the global reward       (1838) */
    RISVEC_EP_OFF_KBIT = 1,        /* traffic/offload_kbit_ep      SUM over the episode           (2047) */
    RISVEC_EP_LOCAL_KBIT = 2,      /* traffic/local_kbit_ep        SUM                            (2048) */
    RISVEC_EP_MEC_CYCLES = 3,      /* queue/mec_cycles             value after the LAST step      (1974) */
    RISVEC_EP_BACKLOG = 4,         /* queue/backlog_kbit_ep_mean                                  (1862) */
    RISVEC_EP_DELAY_LOCAL = 5,     /* delay/local_ep_mean                                         (1858) */
    RISVEC_EP_DELAY_EDGE_Q = 6,    /* delay/edge_queue_ep_mean                                    (1859) */
    RISVEC_EP_DELAY_EDGE_C = 7,    /* delay/edge_compute_ep_mean                                  (1860) */
    RISVEC_EP_DELAY_TX = 8,        /* delay/tx_ep_mean                                            (1861) */
    RISVEC_EP_MEC_UTIL = 9,        /* queue/mec_util_ep_mean                                      (1863) */
    RISVEC_EP_LOCAL_UTIL = 10,     /* cpu/local_util_ep_mean                                      (1864) */
    RISVEC_EP_QOS_VIOL = 11,       /* qos/violation_rate_ep_mean                                  (1865) */
    RISVEC_EP_DELAY = 12,          /* delay/episode_mean  (abs/delay_ms = 1000 x this)            (1850) */
    RISVEC_EP_ENERGY = 13,         /* energy/episode_mean (abs/energy_J)                          (1851) */
    RISVEC_EP_POWER_OFFLOAD = 14,  /* power/offload_avg                                           (1843) */
    RISVEC_EP_POWER_LOCAL = 15,    /* power/local_avg                                             (1842) */
    RISVEC_EP_POWER_TOTAL = 16,    /* power/total_avg                                             (1841) */
    RISVEC_EP_MIN_USER = 17,       /* min over users of the per-user episode mean                 (1939) */
    RISVEC_EP_VAR_USER = 18,       /* population variance of the same                             (1940) */
    RISVEC_EP_JAIN = 19,           /* (sum x)^2 / (V sum x^2 + 1e-12)                         (112-119) */
    RISVEC_EP_BEST_GLOBAL = 20     /* best global reward of the episode                           (1613) */
};

/* Zero acc (column 16 to -inf).  Call where the driver resets its ep_* sums (TRAIN:1278-1300). */
int risvec_episode_clear(int32_t n_envs, int32_t n_veh, double *acc, risvec_stream_t stream);

/* One step's contribution: metrics [E,16] and reward [E,V] as the step kernel wrote them, power_w
 * [E,2,V] or NULL (then columns 14/15 stay untouched, as when the env has no last_power_W). */
int risvec_episode_accumulate(int32_t n_envs, int32_t n_veh, const float *metrics, const float *reward,
                              const float *power_w, float user_clip, double *acc, risvec_stream_t stream);

/* Episode end.  per_env [E, RISVEC_EP_COLS] (optional) receives every env's episode scalars; summary
 * [3, RISVEC_EP_COLS] their mean / min / max over the envs, reduced in a fixed order (deterministic);
 * partial is scratch of risvec_episode_partial_rows(n_envs) x 3 x RISVEC_EP_COLS doubles.  n_steps
 * >= 1; metrics supplies column 3 (the queue after the last step). */
int32_t risvec_episode_partial_rows(int32_t n_envs);
int risvec_episode_summary(int32_t n_envs, int32_t n_veh, int32_t n_steps, const double *acc, const float *metrics,
                           double *per_env, double *partial, double *summary, risvec_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RISVEC_H */
