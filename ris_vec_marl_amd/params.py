"""Environment parameters under the reference's attribute names, and the YAML key map.

`EnvParams` carries the attributes the reference driver pokes on `Environ`
(Environment.py:57-190; marl_train_bcd.py:548-779) and converts them to the POD
`RisVecParams` the kernels take by value.  `apply_yaml_config` reproduces what
`marl_train_bcd.py` does with `config.yaml` for the keys that reach the environment.
"""
from __future__ import annotations

import math
from typing import Any, Dict, Mapping, Optional, Sequence

import numpy as np

from . import _native as N


def reference_lanes() -> Dict[str, list]:
    """Lane coordinates of the reference driver's Config (marl_train_bcd.py:446-449)."""
    up = [i / 2.0 for i in [400 + 3.5 / 2, 400 + 3.5 + 3.5 / 2, 800 + 3.5 / 2, 800 + 3.5 + 3.5 / 2]]
    down = [i / 2.0 for i in [400 - 3.5 - 3.5 / 2, 400 - 3.5 / 2, 800 - 3.5 - 3.5 / 2, 800 - 3.5 / 2]]
    return dict(up_lanes=up, down_lanes=down, left_lanes=list(up), right_lanes=list(down))


def poisson_cdf_table(lam: float) -> np.ndarray:
    """float32 CDF of Poisson(lam), RISVEC_POISSON_TABLE entries, built in float64.
    The kernels draw arrivals (Environment.py:717-719) by counting entries <= u."""
    n = N.POISSON_TABLE
    if not lam > 0:
        return np.ones(n, dtype=np.float32)
    if lam > 16.0:
        raise ValueError("arrival rate %.3g > 16 is outside the %d-entry Poisson table" % (lam, n))
    k = np.arange(n)
    logp = -lam + k * math.log(lam) - np.array([math.lgamma(i + 1) for i in k])
    return np.minimum(np.cumsum(np.exp(logp)), 1.0).astype(np.float32)


CHANNEL_MODELS = {"free": N.CH_FREE, "3gpp_umi": N.CH_3GPP_UMI, "3gpp_uma": N.CH_3GPP_UMA}


class EnvParams:
    """Attribute bag with the reference's names and class defaults (Environment.py:57-190)."""

    _FIELDS = dict(
        channel_model="free", fc_GHz=3.5, bandwidth=1.0, bandwidth_hz=1.0e6, N0_dBm_per_Hz=-174,
        N0_W_per_Hz=10 ** ((-174 - 30) / 10), noise_power=10 ** ((-174 - 30) / 10) * 1.0e6,
        P_max=1.0, power_scale=0.7,
        qos_enable=True, R_min_bpsHz=0.20, D_max_s=0.10, qos_penalty=5.0,
        time_slow=0.1, time_fast=0.001, k=1e-28, L=500,
        f_local_max=1.0e9, f_edge_max=2.0e9, cycles_per_bit=500.0, cpu_share_floor=0.10,
        w_d=0.5, w_e=3.0, reward_clip=50.0, rate=3, data_buf_size=10,
        shadow_std_los=4.0, shadow_std_nlos=7.0, rician_K_dB=0.0, vehAntGain=3,
        # dead on the MARL path but part of the surface (Environment.py:86-97, 118-142)
        Decorrelation_distance=10, sig2_dB=-110, sig2=10 ** (-110 / 10), bsAntGain=8,
        bsNoiseFigure=5, vehNoiseFigure=9, delay_mean=0.0, delay_var=1.0, energy_mean=0.0,
        energy_var=1.0, reward_norm_beta=0.99, sample_weights=True, w_d_range=(0.2, 1.0),
        w_e_range=(2.0, 6.0), w_fair_range=(0.2, 1.0), w_fair=0.5, reward_scale=10.0,
    )

    def __init__(self):
        object.__setattr__(self, "_version", 0)
        for k, v in self._FIELDS.items():
            object.__setattr__(self, k, v)

    def __setattr__(self, key, value):
        object.__setattr__(self, key, value)
        object.__setattr__(self, "_version", self._version + 1)

    @property
    def version(self) -> int:
        return self._version

    def to_c(self, lanes: Mapping[str, Sequence[float]], width: float, height: float) -> N.RisVecParams:
        p = N.RisVecParams()
        p.abi_version = N.ABI_VERSION
        p.struct_bytes = N.C.sizeof(N.RisVecParams)
        p.bandwidth_mhz = float(self.bandwidth)
        p.noise_power = float(self.noise_power)
        p.p_max = float(self.P_max)
        p.power_scale = float(self.power_scale)
        p.qos_enable = 1 if self.qos_enable else 0
        p.r_min_bpshz = float(self.R_min_bpsHz)
        p.d_max_s = float(self.D_max_s)
        p.qos_penalty = float(self.qos_penalty)
        p.time_fast = float(self.time_fast)
        p.k_cpu = float(self.k)
        p.f_local_max = float(self.f_local_max)
        p.f_edge_max = float(self.f_edge_max)
        p.cycles_per_bit = float(self.cycles_per_bit)
        p.cpu_share_floor = float(self.cpu_share_floor)
        p.w_d = float(self.w_d)
        p.w_e = float(self.w_e)
        p.reward_clip = float(self.reward_clip)
        p.arrival_rate = float(self.rate)
        p.poisson_cdf[:] = poisson_cdf_table(float(self.rate)).tolist()
        p.fc_ghz = float(self.fc_GHz)
        p.shadow_std_los = float(self.shadow_std_los)
        p.shadow_std_nlos = float(self.shadow_std_nlos)
        p.rician_k_db = float(self.rician_K_dB)
        p.veh_ant_gain = float(self.vehAntGain)
        n = len(lanes["up_lanes"])
        for key in ("up_lanes", "down_lanes", "left_lanes", "right_lanes"):
            if len(lanes[key]) != n:
                raise ValueError("all four lane lists must have the same length")
        if not 1 <= n <= N.MAX_LANES:
            raise ValueError("between 1 and %d lanes per direction are supported" % N.MAX_LANES)
        p.n_lanes = n
        p.time_slow = float(self.time_slow)
        p.width = float(width)
        p.height = float(height)
        for i in range(n):
            p.lanes_up[i] = float(lanes["up_lanes"][i])
            p.lanes_down[i] = float(lanes["down_lanes"][i])
            p.lanes_left[i] = float(lanes["left_lanes"][i])
            p.lanes_right[i] = float(lanes["right_lanes"][i])
        return p


# defaults of the reference driver's Config that reach the env (marl_train_bcd.py:426-427, 505-508)
DRIVER_DEFAULTS = dict(w_d_fixed=1.0, w_e_fixed=2.0, use_weight_sampling=False, qos_enable=True,
                       qos_R_min_bpsHz=0.15, qos_D_max_s=0.12, qos_penalty=5.0)


def _as_bool(v: Any) -> bool:
    """marl_train_bcd.py:598-604."""
    if isinstance(v, bool):
        return v
    if isinstance(v, (int, float)):
        return v != 0
    if isinstance(v, str):
        return v.strip().lower() in {"true", "ture", "yes", "y", "on", "1"}
    return bool(v)


def load_yaml(path: str) -> dict:
    import yaml
    with open(path, "r", encoding="utf-8") as f:
        return yaml.safe_load(f) or {}


def apply_yaml_config(env: Any, y: Optional[Mapping[str, Any]]) -> None:
    """Apply a parsed `config.yaml` to an env-like object (anything with the reference's
    attribute names: `Environ` facade, `VecEnviron`, `EnvParams`) exactly as the
    reference driver does (marl_train_bcd.py:548-614, 672-673, 750-779).  Unknown keys
    are ignored, as there.  `y=None` applies only the driver's Config defaults."""
    y = dict(y or {})
    cfg = dict(DRIVER_DEFAULTS)
    mec = y.get("mec", {}) or {}
    phy = y.get("phy", {}) or {}
    rew = y.get("reward", {}) or {}
    env_cfg = y.get("env", {}) or {}
    cfg["w_d_fixed"] = float(rew.get("w_d_fixed", cfg["w_d_fixed"]))                # :559
    cfg["w_e_fixed"] = float(rew.get("w_e_fixed", cfg["w_e_fixed"]))                # :560
    if not bool(rew.get("sample", cfg["use_weight_sampling"])):                      # :562-564
        env.w_d = float(cfg["w_d_fixed"])
        env.w_e = float(cfg["w_e_fixed"])
    env.reward_norm_beta = float(rew.get("norm_beta", getattr(env, "reward_norm_beta", 0.9)))   # :567
    env.rate = float(env_cfg.get("rate", env.rate))                                  # :571
    env.cpu_share_floor = float(env_cfg.get("cpu_share_floor", getattr(env, "cpu_share_floor", 0.10)))  # :573
    env.f_local_max = float(mec.get("f_local_max", env.f_local_max))                 # :580
    env.f_edge_max = float(mec.get("f_edge_max", env.f_edge_max))                    # :581
    env.cycles_per_bit = float(mec.get("cycles_per_bit", env.cycles_per_bit))        # :582
    env.k = float(mec.get("k_cpu", env.k))                                           # :583
    env.cpu_share_floor = float(mec.get("cpu_share_floor", getattr(env, "cpu_share_floor", 0.02)))      # :585
    env.P_max = float(phy.get("P_max", env.P_max))                                   # :588
    env.bandwidth = float(phy.get("bandwidth_MHz", env.bandwidth))                   # :589
    env.bandwidth_hz = env.bandwidth * 1e6                                           # :591
    env.noise_power = env.N0_W_per_Hz * env.bandwidth_hz                             # :592
    env.channel_model = str(phy.get("channel_model", env.channel_model))             # :593
    env.fc_GHz = float(phy.get("fc_GHz", env.fc_GHz))                                # :594
    if "power_scale" in y:                                                           # :607-614
        try:
            env.power_scale = float(y.get("power_scale"))
        except Exception:
            pass
    else:
        env.power_scale = float(getattr(env, "power_scale", 0.7))
    cfg["use_weight_sampling"] = bool(rew.get("sample", cfg["use_weight_sampling"]))  # :636
    cfg["qos_enable"] = bool(y.get("qos_enable", cfg["qos_enable"]))                 # :672
    cfg["qos_penalty"] = float(y.get("qos_penalty", cfg["qos_penalty"]))             # :673
    env.qos_enable = bool(cfg["qos_enable"])                                         # :750
    env.R_min_bpsHz = float(cfg["qos_R_min_bpsHz"])                                  # :751
    env.D_max_s = float(cfg["qos_D_max_s"])                                          # :752
    env.qos_penalty = float(cfg["qos_penalty"])                                      # :753
    if not cfg["use_weight_sampling"]:                                               # :771-773
        env.w_d = cfg["w_d_fixed"]
        env.w_e = cfg["w_e_fixed"]
    env.sample_weights = bool(cfg["use_weight_sampling"])                            # :775
