"""ctypes binding of `csrc/librisvec.so` (the C ABI declared in `include/risvec.h`).

The product path has no CPU fallback: if the shared library is missing this module
raises at import of the symbol table, and every compute call goes through HIP.
"""
from __future__ import annotations

import ctypes as C
import os

ABI_VERSION = 15
POISSON_TABLE = 64
MAX_LANES = 8
MAX_VEH = 64
METRICS = 16
PARTNER_SINGLE = -1
PARTNER_NONE = -2
PARTNER_SECOND = 1 << 16

OK, ERR_ARG, ERR_SHAPE, ERR_LAUNCH, ERR_UNSUPPORTED = range(5)
DIR_U, DIR_D, DIR_L, DIR_R = range(4)
CH_FREE, CH_3GPP_UMI, CH_3GPP_UMA, CH_OTHER = range(4)
STEP_METRICS, STEP_POWER_W, STEP_POLICY_ACTION, STEP_OBS, STEP_REUSE_COLSUM = 1, 2, 4, 8, 16
STEP_REUSE_SSUM = 32
STEP_STEER = 64
STEP_REUSE_IDX = 128
STEP_THETA_BY_INDEX = 256
BCD_REUSE_COLSUM, BCD_REUSE_SSUM, BCD_REUSE_IDX, BCD_NO_THETA = 1, 2, 4, 8

METRIC_NAMES = (
    "global_reward", "last_off_kbit_sum", "last_local_kbit_sum", "last_mec_queue_cycles",
    "last_backlog_kbit_mean", "last_delay_local_mean", "last_delay_edge_q_mean",
    "last_delay_edge_c_mean", "last_t_tx_mean", "last_mec_utilization",
    "last_local_util_mean", "last_qos_violation", "last_delay_mean", "last_energy_mean",
)

_LANES = C.c_double * MAX_LANES


class RisVecParams(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("struct_bytes", C.c_uint32),
        ("bandwidth_mhz", C.c_float), ("noise_power", C.c_float), ("p_max", C.c_float),
        ("power_scale", C.c_float),
        ("qos_enable", C.c_int32), ("r_min_bpshz", C.c_float), ("d_max_s", C.c_float),
        ("qos_penalty", C.c_float),
        ("time_fast", C.c_float), ("k_cpu", C.c_float), ("f_local_max", C.c_float),
        ("f_edge_max", C.c_float), ("cycles_per_bit", C.c_float), ("cpu_share_floor", C.c_float),
        ("w_d", C.c_float), ("w_e", C.c_float), ("reward_clip", C.c_float),
        ("arrival_rate", C.c_float), ("poisson_cdf", C.c_float * POISSON_TABLE),
        ("fc_ghz", C.c_float), ("shadow_std_los", C.c_float), ("shadow_std_nlos", C.c_float),
        ("rician_k_db", C.c_float), ("veh_ant_gain", C.c_float), ("n_lanes", C.c_int32),
        ("time_slow", C.c_double), ("width", C.c_double), ("height", C.c_double),
        ("lanes_up", _LANES), ("lanes_down", _LANES), ("lanes_left", _LANES), ("lanes_right", _LANES),
    ]


EP_FIXED = 17     # RISVEC_EP_FIXED: accumulator columns besides the V per-user sums
EP_COLS = 21      # RISVEC_EP_COLS

_FP = C.c_void_p


class RisVecState(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("struct_bytes", C.c_uint32),
        ("n_envs", C.c_int32), ("n_veh", C.c_int32), ("n_ris", C.c_int32), ("control_bit", C.c_int32),
        ("env_offset", C.c_int64),
        ("pos", _FP), ("dir", _FP), ("vel", _FP),
        ("dist_r", _FP), ("ang_r", _FP), ("pl", _FP), ("h_r", _FP),
        ("theta", _FP), ("b", _FP), ("h_d", _FP), ("gain", _FP),
        ("data_buf", _FP), ("mec_q", _FP),
        ("rate", _FP), ("data_t", _FP), ("data_p", _FP), ("reward", _FP), ("over_power", _FP),
        ("obs", _FP), ("metrics", _FP), ("power_w", _FP), ("c_col", _FP), ("s_sum", _FP), ("over_data", _FP), ("z_r", _FP),
        ("theta_idx", _FP),
    ]


class RisVecTraj(C.Structure):
    _fields_ = [("reward", _FP), ("obs", _FP), ("metrics", _FP)]


class RisVecSarlParams(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("struct_bytes", C.c_uint32),
        ("time_fast", C.c_float), ("bandwidth_mhz", C.c_float), ("k_cpu", C.c_float), ("cycles_l", C.c_float),
        ("t_factor1", C.c_float), ("t_factor2", C.c_float), ("penalty1", C.c_float), ("penalty2", C.c_float),
        ("arrival_rate", C.c_float), ("poisson_cdf", C.c_float * POISSON_TABLE),
    ]


class RisVecNomaParams(C.Structure):
    _fields_ = [
        ("min_pair_target", C.c_int32), ("mwm_backoff_rounds", C.c_int32), ("mwm_allow_singles", C.c_int32),
        ("qos_enable", C.c_int32), ("relax_topk_step", C.c_int32), ("freeze_group_in_episode", C.c_int32),
        ("freeze_recalc_every", C.c_int32), ("mask_enable", C.c_int32),
        ("mwm_accept_quantile", C.c_double), ("mwm_accept_q_step", C.c_double),
        ("completion_min_quantile", C.c_double), ("score_w_delta_db", C.c_double), ("abs_gain_min_db", C.c_double),
        ("qos_soft_penalty", C.c_double), ("qos_R_min", C.c_double), ("noise_power", C.c_double),
        ("P_max", C.c_double), ("relax_tau_factor", C.c_double), ("tau_back_floor_db", C.c_double),
        ("freeze_reward_drop_ratio", C.c_double), ("freeze_unstick_prob", C.c_double),
        ("score_w_history", C.c_float), ("pair_hist_decay", C.c_float),
    ]


class RisVecNomaState(C.Structure):
    _fields_ = [
        ("n_envs", C.c_int32), ("n_veh", C.c_int32), ("env_offset", C.c_int64),
        ("hist", _FP), ("streak", _FP), ("partner", _FP), ("n_groups", _FP), ("last_global", _FP),
        ("best_global", _FP), ("flags", _FP), ("mask", _FP), ("tau", _FP), ("pending", _FP),
        ("scratch", _FP), ("scratch_bytes", C.c_int64),
    ]


class RisVecReplay(C.Structure):
    _fields_ = [
        ("n_agents", C.c_int32), ("input_shape", C.c_int32), ("n_actions", C.c_int32), ("reserved", C.c_int32),
        ("mem_size", C.c_int64),
        ("state_memory", _FP), ("action_memory", _FP), ("reward_global_memory", _FP), ("reward_local_memory", _FP),
        ("new_state_memory", _FP), ("terminal_memory", _FP), ("mask_memory", _FP),
    ]


class RisVecStepRing(C.Structure):
    _fields_ = [("rb", RisVecReplay), ("mem_cntr", C.c_int64), ("probs", _FP), ("mask", _FP), ("done", C.c_int32),
                ("reserved", C.c_int32)]


NOMA_MAX_VEH = 16
NOMA_HAS_LAST, NOMA_UNSTICK_USED, NOMA_HAS_GROUPS = 1, 2, 4

LIB_PATH = os.environ.get("RISVEC_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "librisvec.so")
# (RISVEC_LIB: an alternative build of the same library, for A/B experiments)

_PROTOS = {
    "risvec_abi_version": (C.c_uint32, []),
    "risvec_last_error": (C.c_char_p, []),
    "risvec_last_kernel": (C.c_char_p, []),
    "risvec_default_params": (None, [C.POINTER(RisVecParams)]),
    "risvec_reset": (C.c_int, [C.POINTER(RisVecState), C.POINTER(RisVecParams), _FP, _FP,
                               C.c_uint64, C.c_uint32, _FP]),
    "risvec_mobility": (C.c_int, [C.POINTER(RisVecState), C.POINTER(RisVecParams), _FP, _FP,
                                  C.c_uint64, C.c_uint32, _FP]),
    "risvec_geometry": (C.c_int, [C.POINTER(RisVecState), C.POINTER(RisVecParams), _FP]),
    "risvec_gain": (C.c_int, [C.POINTER(RisVecState), C.POINTER(RisVecParams), _FP]),
    "risvec_gain_3gpp": (C.c_int, [C.POINTER(RisVecState), C.POINTER(RisVecParams), C.c_int32,
                                   _FP, _FP, _FP, C.c_uint64, C.c_uint32, _FP]),
    "risvec_colsum": (C.c_int, [C.POINTER(RisVecState), _FP]),
    "risvec_bcd": (C.c_int, [C.POINTER(RisVecState), C.POINTER(RisVecParams), _FP, C.c_uint32, _FP]),
    "risvec_theta_from_index": (C.c_int, [C.POINTER(RisVecState), _FP]),
    "risvec_theta_by_index_supported": (C.c_int, [C.c_int32, C.c_int32]),
    "risvec_set_phase": (C.c_int, [C.POINTER(RisVecState), _FP, _FP]),
    "risvec_random_phase": (C.c_int, [C.POINTER(RisVecState), _FP, C.c_uint64, C.c_uint32, _FP]),
    "risvec_step": (C.c_int, [C.POINTER(RisVecState), C.POINTER(RisVecParams), _FP, _FP, _FP, _FP,
                              C.c_uint64, C.c_uint32, C.c_uint32, _FP]),
    "risvec_data_rate": (C.c_int, [C.POINTER(RisVecState), C.POINTER(RisVecParams), _FP, _FP, _FP, _FP, _FP]),
    "risvec_step_fused": (C.c_int, [C.POINTER(RisVecState), C.POINTER(RisVecParams), _FP, _FP, _FP,
                                    _FP, C.c_uint64, C.c_uint32, C.c_uint32, _FP]),
    "risvec_step_fused_multi": (C.c_int, [C.POINTER(RisVecState), C.POINTER(RisVecParams), C.c_int32, _FP, _FP, _FP, _FP,
                                          C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(RisVecTraj), _FP]),
    "risvec_step_multi": (C.c_int, [C.POINTER(RisVecState), C.POINTER(RisVecParams), C.c_int32, _FP, _FP, _FP, _FP,
                                    C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(RisVecTraj), _FP]),
    "risvec_sarl_step": (C.c_int, [C.POINTER(RisVecState), C.POINTER(RisVecSarlParams), _FP, _FP, _FP,
                                   C.c_uint64, C.c_uint32, C.c_uint32, _FP]),
    "risvec_step_fused_bcd": (C.c_int, [C.POINTER(RisVecState), C.POINTER(RisVecParams), _FP, _FP,
                                        _FP, _FP, C.c_uint64, C.c_uint32, C.c_uint32, _FP]),
    "risvec_step_ring": (C.c_int, [C.POINTER(RisVecState), C.POINTER(RisVecParams), C.POINTER(RisVecStepRing), _FP, _FP, _FP,
                                   _FP, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int32, _FP]),
    "risvec_noma_default_params": (None, [C.POINTER(RisVecNomaParams), C.c_int32]),
    "risvec_noma_begin_episode": (C.c_int, [C.POINTER(RisVecNomaState), _FP]),
    "risvec_noma_mask": (C.c_int, [C.POINTER(RisVecNomaState), _FP, _FP, C.c_double, C.c_int32, _FP]),
    "risvec_noma_group": (C.c_int, [C.POINTER(RisVecNomaState), C.POINTER(RisVecNomaParams), _FP, _FP, _FP,
                                    C.c_int32, C.c_int32, _FP, _FP, C.c_int32, C.c_int32, _FP,
                                    C.c_uint64, C.c_uint32, _FP, _FP]),
    "risvec_noma_group_raw": (C.c_int, [C.POINTER(RisVecNomaState), C.POINTER(RisVecNomaParams), _FP, _FP, _FP,
                                        C.c_int32, C.c_int32, _FP, _FP, C.c_int32, C.c_int32, _FP,
                                        C.c_uint64, C.c_uint32, _FP, _FP]),
    "risvec_noma_flush": (C.c_int, [C.POINTER(RisVecNomaState), C.POINTER(RisVecNomaParams), _FP]),
    "risvec_noma_scratch_bytes": (C.c_int64, [C.c_int32, C.c_int32]),
    "risvec_replay_store": (C.c_int, [C.POINTER(RisVecReplay), C.c_int64, C.c_int32, _FP, _FP, _FP, C.c_int32, _FP,
                                      _FP, _FP, C.c_int32, _FP, _FP, _FP]),
    "risvec_replay_store_policy": (C.c_int, [C.POINTER(RisVecReplay), C.c_int64, C.c_int32, _FP, _FP, _FP, _FP, C.c_int32, _FP,
                                             _FP, _FP, C.c_int32, _FP, _FP, _FP]),
    "risvec_replay_sample": (C.c_int, [C.POINTER(RisVecReplay), C.c_int64, C.c_int32, _FP, C.c_uint64, C.c_uint32,
                                       _FP, _FP, _FP, _FP, _FP, _FP, _FP, _FP, _FP]),
    "risvec_marshal_actions": (C.c_int, [C.c_int32, C.c_int32, _FP, _FP, C.c_float, _FP, _FP, _FP, _FP]),
    "risvec_policy_layer1": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, _FP, _FP, _FP, _FP, _FP, _FP, _FP]),
    "risvec_policy_layer1_split16": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, _FP, _FP, _FP, _FP, _FP, _FP, _FP]),
    "risvec_policy_mlp_supported": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "risvec_policy_mlp": (C.c_int, [C.c_int32] * 6 + [_FP] * 13),
    "risvec_policy_heads": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, _FP, _FP, _FP, _FP, _FP, _FP, _FP, _FP]),
    "risvec_policy_sample": (C.c_int, [C.c_int32, C.c_int32, C.c_int64, _FP, _FP, _FP, _FP, _FP, _FP, C.c_uint64, C.c_uint32,
                                       C.c_float, _FP, _FP, _FP, _FP, _FP, _FP, _FP]),
    "risvec_episode_clear": (C.c_int, [C.c_int32, C.c_int32, _FP, _FP]),
    "risvec_episode_accumulate": (C.c_int, [C.c_int32, C.c_int32, _FP, _FP, _FP, C.c_float, _FP, _FP]),
    "risvec_episode_partial_rows": (C.c_int32, [C.c_int32]),
    "risvec_episode_summary": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _FP, _FP, _FP, _FP, _FP, _FP]),
}

EXPORTS = tuple(_PROTOS)   # every symbol include/risvec.h declares

_lib = None


def load():
    """Load librisvec.so (once) and return the ctypes handle; raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(
            "ris_vec_marl_amd: HIP extension not built (%s missing). Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C ris_vec_marl_amd/csrc`. "
            "There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    got = lib.risvec_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError("librisvec.so ABI %d != binding ABI %d (rebuild)" % (got, ABI_VERSION))
    if C.sizeof(RisVecParams) != _sizeof_check(lib):
        raise RuntimeError("RisVecParams layout mismatch between ctypes and librisvec.so")
    _lib = lib
    return lib


def _sizeof_check(lib) -> int:
    p = RisVecParams()
    lib.risvec_default_params(C.byref(p))
    return int(p.struct_bytes)


def resolve_device(device):
    """`torch.device(device)` with the index of a bare "cuda" filled in (the current device), so that
    `tensor.device == self.device` holds for tensors the object allocates itself: torch compares
    `cuda` and `cuda:0` as different devices.  Without a HIP device the argument is returned as
    given (the compute calls raise later: there is no CPU fallback)."""
    import torch
    d = torch.device(device if device is not None else "cuda")
    if d.type == "cuda" and d.index is None and torch.cuda.is_available():
        d = torch.device("cuda", torch.cuda.current_device())
    return d


def last_kernel() -> str:
    """Name of the kernel the calling thread's last step / BCD call dispatched (risvec_last_kernel)."""
    k = load().risvec_last_kernel()
    return k.decode() if k else ""


class RisVecError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("librisvec error %d: %s" % (code, msg))
        self.code = code


def check(rc: int) -> None:
    if rc != OK:
        msg = load().risvec_last_error()
        err = RisVecError(rc, msg.decode() if msg else "?")
        if rc in (ERR_ARG, ERR_SHAPE):
            raise ValueError(str(err)) from err
        raise err
