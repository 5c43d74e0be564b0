"""f1: the single-agent environment variant, `Simulation-SARL/Environment.py`
(SENV below) - the env `ddpg_torch.py` drives.  Its geometry, mobility, reset and RIS cascade are the
MARL ones; only `step(action_power, action_phase)` differs (SENV:321-359): the phases come from
the agent, the rate is a natural log against sigma^2, local processing follows the cube-root CPU
model, the reward is power + buffer length with two penalties.

`SarlEnviron` is the E=1 facade with the reference's constructor and 6-tuple `step`; batched use
goes through `VecEnviron.sarl_step`.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import _native as N
from .compat import Environ
from .params import poisson_cdf_table


class SarlParams:
    """SENV:66-83 class defaults under the reference's attribute names."""

    def __init__(self):
        self.time_fast = 0.001      # SENV:67
        self.bandwidth = 1          # SENV:69 (MHz)
        self.k = 1e-28              # SENV:70
        self.L = 500                # SENV:71
        self.rate = 3               # SENV:78
        self.t_factor1 = 1          # SENV:80
        self.t_factor2 = 0.6        # SENV:81
        self.penalty1 = 2           # SENV:82
        self.penalty2 = 2           # SENV:83

    def to_c(self) -> N.RisVecSarlParams:
        p = N.RisVecSarlParams()
        p.abi_version = N.ABI_VERSION
        p.struct_bytes = N.C.sizeof(N.RisVecSarlParams)
        p.time_fast, p.bandwidth_mhz, p.k_cpu, p.cycles_l = float(self.time_fast), float(self.bandwidth), float(self.k), float(self.L)
        p.t_factor1, p.t_factor2 = float(self.t_factor1), float(self.t_factor2)
        p.penalty1, p.penalty2 = float(self.penalty1), float(self.penalty2)
        p.arrival_rate = float(self.rate)
        p.poisson_cdf[:] = poisson_cdf_table(float(self.rate)).tolist()
        return p


def sarl_action_map(action: torch.Tensor, n_veh: int, M: int):
    """ddpg_train.py:149-158: agent output [E, 2V+M] in [-1,1] -> (action_power [E,2,V],
    action_phase [E,M] in [0, 2 pi)).  Pure data marshalling of the policy output."""
    a = action.clamp(-0.999, 0.999)
    power = torch.stack([(a[:, :n_veh] + 1) / 2, (a[:, n_veh:2 * n_veh] + 1) / 2], dim=1)
    phase = ((a[:, 2 * n_veh:2 * n_veh + M] + 1) / 2) * (math.pi * 2)
    return power.contiguous(), phase.contiguous()


def sarl_observe(env, action_phase: torch.Tensor) -> torch.Tensor:
    """ddpg_train.py:47-73 for all agents: [E, V, M//V + 5] = each agent's slice of the phase
    action followed by the 5-float tail the step kernel wrote into `obs`."""
    E, V = env.n_envs, env.n_veh
    tn = env.M // V
    th = action_phase[:, :tn * V].reshape(E, V, tn)
    return torch.cat([th, env.tensors["obs"]], dim=2)


class SarlEnviron(Environ):
    """`Simulation-SARL/Environment.py:Environ` surface over one device-resident env."""

    def __init__(self, down_lane, up_lane, left_lane, right_lane, width, height, n_veh, M, control_bit,
                 device: str = "cuda", seed: int = 0):
        super().__init__(down_lane, up_lane, left_lane, right_lane, width, height, n_veh, M, control_bit,
                         device=device, seed=seed)
        object.__setattr__(self, "sarl", SarlParams())
        self.Reward = 0.0

    # the SARL attribute names that differ from / shadow the MARL parameter bag
    def __getattr__(self, name):
        sp = self.__dict__.get("sarl")
        if sp is not None and name in ("t_factor1", "t_factor2", "penalty1", "penalty2"):
            return getattr(sp, name)
        return super().__getattr__(name)

    def __setattr__(self, name, value):
        sp = self.__dict__.get("sarl")
        if sp is not None and name in ("t_factor1", "t_factor2", "penalty1", "penalty2", "rate", "k", "L",
                                       "bandwidth", "time_fast"):
            setattr(sp, name, value)
            if name in ("t_factor1", "t_factor2", "penalty1", "penalty2"):
                return
        super().__setattr__(name, value)

    def _sarl_launch(self, a: np.ndarray, ph: np.ndarray, arrivals) -> None:
        """One env, one step: power | phase | arrivals go to the device in ONE copy of pinned 32-bit words and the launch
        is pre-bound (re-bound when a parameter of `self.sarl` changed), as in `Environ._step_launch`."""
        import torch
        V, M, vec = self.n_veh, self.M, self._vec
        key = tuple(sorted(vars(self.sarl).items()))
        st = self.__dict__.get("_sarl_stage")
        if st is None or st["key"] != key:
            vec._ensure_device()
            up4 = lambda n: (n + 3) // 4 * 4                   # noqa: E731  (16-byte aligned sections)
            o_ph = up4(2 * V)
            o_ar = o_ph + up4(M)
            host = torch.zeros(o_ar + up4(V), dtype=torch.int32).pin_memory()
            dev = torch.zeros(o_ar + up4(V), dtype=torch.int32, device=vec.device)
            hn = host.numpy()
            d_a = dev[:2 * V].view(torch.float32).view(1, 2, V)
            d_ph = dev[o_ph:o_ph + M].view(torch.float32).view(1, M)
            d_ar = dev[o_ar:o_ar + V].view(1, V)
            st = dict(key=key, host=host, dev=dev,
                      a=hn[:2 * V].view(np.float32), ph=hn[o_ph:o_ph + M].view(np.float32), ar=hn[o_ar:o_ar + V],
                      plain=vec.bind_sarl_step(d_a, d_ph, None, sarl_params=self.sarl),
                      injected=vec.bind_sarl_step(d_a, d_ph, d_ar, sarl_params=self.sarl))
            object.__setattr__(self, "_sarl_stage", st)
        st["a"][:] = a.reshape(-1)
        st["ph"][:] = ph
        if arrivals is not None:
            arr = np.asarray(arrivals)
            if arr.shape != (V,):
                raise ValueError("arrivals must have shape [n_veh]")
            st["ar"][:] = arr
        st["dev"].copy_(st["host"], non_blocking=True)
        st["injected" if arrivals is not None else "plain"]()

    def step(self, action_power, action_phase, arrivals=None):   # noqa: D102  (signature of SENV:321)
        a = np.asarray(action_power, dtype=np.float64)
        ph = np.asarray(action_phase, dtype=np.float64)
        if a.shape != (2, self.n_veh) or ph.shape != (self.M,):
            raise ValueError("step(action_power [2,n_veh], action_phase [M])")
        self.elements_phase_shift_real = action_phase
        self._sarl_launch(a, ph, arrivals)
        self._dirty()
        self.Reward = float(self._host("metrics")[0])
        return (self.Reward, self.DataBuf, self.data_t, self.data_p, self._host("over_power"), self.over_data)
