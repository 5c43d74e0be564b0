"""Drop-in facade: the reference's `Environ` class surface over ONE device-resident env.

`Environ(down_lane, up_lane, left_lane, right_lane, width, height, n_veh, M, control_bit)`
has the constructor, the 16 methods and the attribute names of
`Simulation-MARL-BCD/Environment.py:56`, takes and returns NumPy
float64 like the reference, and runs every computation through the HIP kernels
(E = 1 view of `VecEnviron`).  It exists so the reference's driver and agents
(`marl_train_bcd.py`, `sac_agent.py`, `global_sac_critic.py`) can run unchanged; for
throughput use `VecEnviron` directly.

Differences a maintainer must know (also in INTEGRATION.md):
  * random draws come from a counter-based Philox stream seeded by `seed=`, not from
    the global `numpy.random` state (`set_seed`, marl_train_bcd.py:32-42, has no effect);
  * returned arrays are host copies, not live aliases of internal state: read right after the call (as the
    driver does, marl_train_bcd.py:1611-1662) they hold the reference's values; the reference's own tuple
    keeps changing under the next step (tests/golden/facade_alias_8.npz records how);
  * `noma_groups` may list a vehicle several times, hold pairs [u, u], groups of other sizes and empty groups:
    the reference's semantics are reproduced (the last 1- or 2-element group listing a vehicle decides its rate,
    a partner paired with it earlier keeps the pair rate, every group counts in G; tests/golden/facade_groups_8.npz).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import _native as N
from .params import EnvParams
from .vec_env import ParamAttrs, VecEnviron

_DIR_CHARS = "udlr"


def encode_noma_groups(groups_per_env: Sequence[Sequence[Sequence[int]]], n_veh: int
                       ) -> Tuple[np.ndarray, np.ndarray]:
    """list (per env) of `noma_groups` lists-of-lists (Environment.py:330, 339-369) ->
    (partner [E,V] int32, n_groups [E] int32).

    partner[v] = j if v is listed first in the pair [v, j]; j + 65536 if listed second;
    -1 if v is alone in a 1-element group; -2 if v is in no group or in a group of any
    other size (rate 0, Environment.py:336, 344, 351).  n_groups = len(noma_groups),
    including groups of other sizes, as in Environment.py:341.

    The reference walks the list in order and every group OVERWRITES the rates of its members
    (Environment.py:344-369), so a vehicle listed more than once ends up with the rate of the LAST 1- or
    2-element group that lists it, while a vehicle that was paired with it earlier keeps the rate it got in
    that pair.  The encoding is per vehicle (its own partner entry is all a lane reads), so "last writer wins"
    per entry reproduces exactly that; a pair [u, u] degenerates to the OMA rate, as it does in the reference."""
    E = len(groups_per_env)
    partner = np.full((E, n_veh), N.PARTNER_NONE, dtype=np.int32)
    n_groups = np.zeros(E, dtype=np.int32)
    for e, groups in enumerate(groups_per_env):
        n_groups[e] = len(groups)
        for g in groups:
            if len(g) not in (1, 2):
                continue
            for u in g:
                if not 0 <= int(u) < n_veh:
                    raise ValueError("noma_groups: vehicle index %d outside [0, %d)" % (int(u), n_veh))
            if len(g) == 1:
                partner[e, int(g[0])] = N.PARTNER_SINGLE
            else:
                partner[e, int(g[0])] = int(g[1])
                partner[e, int(g[1])] = int(g[0]) + N.PARTNER_SECOND
    return partner, n_groups


class Vehicle:
    """Environment.py:45-53."""

    def __init__(self, start_position, start_direction, velocity):
        self.position = start_position
        self.direction = start_direction
        self.velocity = velocity
        self.neighbors = []
        self.destinations = []


class Environ(ParamAttrs):
    def __init__(self, down_lane, up_lane, left_lane, right_lane, width, height, n_veh, M, control_bit,
                 device: str = "cuda", seed: int = 0):
        object.__setattr__(self, "params", EnvParams())
        self._vec = VecEnviron(down_lane, up_lane, left_lane, right_lane, width, height, n_veh, M,
                               control_bit, n_envs=1, device=device, seed=seed, params=self.params)
        v = self._vec
        self.down_lanes, self.up_lanes = v.down_lanes, v.up_lanes
        self.left_lanes, self.right_lanes = v.left_lanes, v.right_lanes
        self.width, self.height = width, height
        self.n_veh, self.M, self.control_bit = v.n_veh, v.M, v.control_bit
        self.possible_angles = v.possible_angles
        self.distance_B_R, self.angle_B_R = v.distance_B_R, v.angle_B_R
        self.phase_R = v.phase_R.copy()
        self.elements_phase_shift_real = np.zeros(self.M)
        # legacy / dead attributes kept for surface parity (Environment.py:86-90, 157)
        self.V2I_Shadowing = np.zeros(self.n_veh)
        self.V2I_pathloss = np.zeros(self.n_veh)
        self.V2I_channels_abs = np.zeros(self.n_veh)
        self.delta_distance = []
        self.data_r = np.zeros(self.n_veh)
        # exist from __init__ in the reference (Environment.py:145-147)
        self.last_off_kbit_sum = 0.0
        self.last_local_kbit_sum = 0.0
        self.last_mec_queue_cycles = 0.0
        self._n_vehicles = 0
        self._cache = {}
        self._stage = None       # step(): pinned host words, their device mirror and the two pre-bound launchers

    # ---------------------------------------------------------------- host <-> device plumbing
    def _host(self, key: str) -> np.ndarray:
        if key not in self._cache:
            v = self._vec
            v._ensure_device()
            if key in v._out_offsets:                 # step()'s state / outputs: ONE device-to-host copy serves them all
                if "_slab" not in self._cache:
                    self._cache["_slab"] = v._out_slab.cpu().numpy()
                off, shp = v._out_offsets[key]
                n = int(np.prod(shp))
                self._cache[key] = self._cache["_slab"][off:off + n].reshape(shp)[0].astype(np.float64)
            else:
                self._cache[key] = v.tensors[key][0].detach().cpu().numpy().astype(np.float64)
        return self._cache[key]

    def _upload(self, key: str, value) -> None:
        t = self._vec.tensors[key]
        t[0].copy_(torch.as_tensor(np.asarray(value), dtype=t.dtype).reshape(t[0].shape))
        self._cache.clear()

    def _dirty(self) -> None:
        self._cache.clear()

    DataBuf = property(lambda s: s._host("data_buf"), lambda s, v: s._upload("data_buf", v))
    data_t = property(lambda s: s._host("data_t"), lambda s, v: s._upload("data_t", v))
    data_p = property(lambda s: s._host("data_p"), lambda s, v: s._upload("data_p", v))
    over_data = property(lambda s: s._host("over_data"))
    vehicle_rate = property(lambda s: s._host("rate"), lambda s, v: s._upload("rate", v))
    channel_gains = property(lambda s: s._host("gain"), lambda s, v: s._upload("gain", v))
    distances_R_i = property(lambda s: s._host("dist_r"))
    angles_R_i = property(lambda s: s._host("ang_r"))

    @property
    def mec_queue_cycles(self) -> float:
        return float(self._host("mec_q"))

    @mec_queue_cycles.setter
    def mec_queue_cycles(self, value) -> None:
        self._upload("mec_q", float(value))

    @property
    def elements_phase_shift_complex(self) -> np.ndarray:
        t = self._host("theta")
        return t[:, 0] + 1j * t[:, 1]

    @elements_phase_shift_complex.setter
    def elements_phase_shift_complex(self, value) -> None:
        z = np.asarray(value, dtype=np.complex128).reshape(self.M)
        self._upload("theta", np.stack([z.real, z.imag], -1))
        self._vec._theta_changed()            # the cached sum theta.c and the candidate indices are stale

    @property
    def phases_R_i(self) -> np.ndarray:
        t = self._host("h_r")
        return t[..., 0] + 1j * t[..., 1]

    @property
    def vehicles(self) -> List[Vehicle]:
        pos = self._host("pos")
        dr = self._vec.tensors["dir"][0].cpu().numpy()
        vel = self._host("vel")
        return [Vehicle([float(pos[i, 0]), float(pos[i, 1])], _DIR_CHARS[int(dr[i])], int(vel[i]))
                for i in range(self._n_vehicles)]

    @vehicles.setter
    def vehicles(self, value) -> None:
        value = list(value)
        if len(value) > self.n_veh:
            raise ValueError("at most n_veh vehicles")
        self._n_vehicles = 0
        for veh in value:
            self.add_new_vehicles(veh.position, veh.direction, veh.velocity)

    # ---------------------------------------------------------------- reference methods
    def add_new_vehicles(self, start_position, start_direction, start_velocity) -> None:
        """Environment.py:378-379."""
        i = self._n_vehicles
        if i >= self.n_veh:
            raise ValueError("the device state holds exactly n_veh=%d vehicles" % self.n_veh)
        t = self._vec.tensors
        t["pos"][0, i] = torch.tensor([float(start_position[0]), float(start_position[1])], dtype=torch.float64)
        t["dir"][0, i] = _DIR_CHARS.index(start_direction)
        t["vel"][0, i] = float(start_velocity)
        self._n_vehicles = i + 1
        self._dirty()

    def add_new_vehicles_by_number(self, n) -> None:
        """Environment.py:381-410.  The device spawn kernel always fills the n_veh slots with
        n_veh//4 rounds (+ n_veh%4 extras), which is the only way the reference calls it.
        Like the reference it leaves DataBuf alone."""
        if int(n) != self.n_veh // 4:
            raise ValueError("add_new_vehicles_by_number(n) is supported for n == n_veh//4 "
                             "(as make_new_game calls it, Environment.py:735)")
        keep = self._vec.tensors["data_buf"].clone()
        self._vec.make_new_game()
        self._vec.tensors["data_buf"].copy_(keep)
        self._n_vehicles = self.n_veh
        self._dirty()

    def make_new_game(self, spawn_ints=None, buf0=None) -> None:
        """Environment.py:733-737.  spawn_ints [V,3] / buf0 (scalar) inject the reference's
        own draws (parity tests); default: Philox."""
        if (spawn_ints is None) != (buf0 is None):
            raise ValueError("spawn_ints and buf0 must be given together")
        si = None if spawn_ints is None else np.asarray(spawn_ints)[None]
        b0 = None if buf0 is None else np.asarray(buf0).reshape(1)
        self._vec.make_new_game(si, b0)
        self._n_vehicles = self.n_veh
        self._dirty()

    def renew_positions(self, u_turn=None) -> None:
        """Environment.py:412-542."""
        self._vec.renew_positions(None if u_turn is None else np.asarray(u_turn)[None])
        self._dirty()

    def compute_parms(self) -> None:
        """Environment.py:241-253."""
        self._vec.compute_parms()
        self._dirty()

    def optimize_phase_shift(self) -> None:
        """Environment.py:208-220."""
        self._vec.optimize_phase_shift()
        self._dirty()

    def optimize_compute_objective_function(self) -> float:
        """Environment.py:222-231: sum_v |ro img / (sqrt(d_v^a1) sqrt(dBR^a2))|^2 / sigma^2 with
        img = the sum over the WHOLE [V,M] product.  Evaluated on the device tensors."""
        from .vec_env import sigma
        t = self._vec.tensors
        th = torch.view_as_complex(t["theta"]).to(torch.complex128)[0]
        hr = torch.view_as_complex(t["h_r"]).to(torch.complex128)[0]
        b = torch.view_as_complex(t["b"]).to(torch.complex128)
        img = (th[None, :] * hr * b[None, :]).sum()
        return float((t["pl"][0].double() * (img.abs() ** 2)).sum() / sigma ** 2)

    def update_channel_gains(self, u_los=None, z_shadow=None, small=None) -> None:
        """Environment.py:255-327."""
        w = lambda x: None if x is None else np.asarray(x)[None]   # noqa: E731
        self._vec.update_channel_gains(w(u_los), w(z_shadow), w(small))
        self._dirty()

    def get_channel_gains(self) -> np.ndarray:
        """Environment.py:374-376."""
        return self.channel_gains

    def Random_phase(self, idx=None) -> None:
        """Environment.py:203-206."""
        self._vec.Random_phase(None if idx is None else np.asarray(idx)[None])
        self._dirty()
        th = self.elements_phase_shift_complex
        self.elements_phase_shift_real = list(np.mod(np.angle(th), 2 * np.pi))

    def get_next_phase(self, action_phase) -> None:
        """Environment.py:233-239."""
        self.elements_phase_shift_real = action_phase
        self._vec.get_next_phase(np.asarray(action_phase, dtype=np.float64)[None])
        self._dirty()

    def compute_data_rate(self, power, noma_groups) -> np.ndarray:
        """Environment.py:331-372 (power [2,V] in W; row 0 is the offload power)."""
        partner, ng = encode_noma_groups([noma_groups], self.n_veh)
        r = self._vec.data_rate(np.asarray(power, dtype=np.float64)[None, 0, :], partner, ng)
        return r[0].detach().cpu().numpy().astype(np.float64)

    def step(self, action_power, noma_groups, arrivals=None):
        """Environment.py:547-731.  Returns (per_user_reward, global_reward, DataBuf, data_t,
        data_p, over_power, over_data)."""
        a = np.asarray(action_power, dtype=np.float64)
        if a.shape != (2, self.n_veh):
            raise ValueError("action_power must have shape [2, n_veh]")
        partner, ng = encode_noma_groups([noma_groups], self.n_veh)
        self._step_launch(a, partner[0], int(ng[0]), arrivals)
        self._dirty()
        m = self._host("metrics")
        for i, name in enumerate(N.METRIC_NAMES[1:], start=1):
            if name == "last_qos_violation" and not self.params.qos_enable:
                continue                                            # Environment.py:670-677
            object.__setattr__(self, name, float(m[i]))
        self.last_power_W = self._host("power_w")
        return (self._host("reward"), float(m[0]), self.DataBuf, self.data_t, self.data_p,
                self._host("over_power"), self.over_data)

    def _step_launch(self, a: np.ndarray, partner: np.ndarray, ng: int, arrivals) -> None:
        """One env, one step: the inputs go to the device in ONE copy (a pinned staging buffer of 32-bit words:
        action | partner | n_groups | arrivals) and the launch is pre-bound -- the per-call cost of the facade is what
        a script that swaps `Environment` for this module pays on every step."""
        V, vec = self.n_veh, self._vec
        if self._stage is None:
            vec._ensure_device()
            up4 = lambda n: (n + 3) // 4 * 4                   # noqa: E731  (the C ABI wants 16-byte aligned pointers)
            o_p = up4(2 * V)
            o_g = o_p + up4(V)
            o_r = o_g + 4
            host = torch.zeros(o_r + up4(V), dtype=torch.int32).pin_memory()
            dev = torch.zeros(o_r + up4(V), dtype=torch.int32, device=vec.device)
            hn = host.numpy()
            views = dict(a=hn[:2 * V].view(np.float32), p=hn[o_p:o_p + V], g=hn[o_g:o_g + 1], r=hn[o_r:o_r + V])
            d_a = dev[:2 * V].view(torch.float32).view(1, 2, V)
            d_p, d_g, d_r = dev[o_p:o_p + V].view(1, V), dev[o_g:o_g + 1], dev[o_r:o_r + V].view(1, V)
            self._stage = dict(host=host, dev=dev, views=views,
                               plain=vec.bind_step(d_a, d_p, d_g, None, fused=False),
                               injected=vec.bind_step(d_a, d_p, d_g, d_r, fused=False))
        st = self._stage
        v = st["views"]
        v["a"][:] = a.reshape(-1)
        v["p"][:] = partner
        v["g"][0] = ng
        if arrivals is not None:
            arr = np.asarray(arrivals)
            if arr.shape != (V,):
                raise ValueError("arrivals must have shape [n_veh]")
            v["r"][:] = arr
        st["dev"].copy_(st["host"], non_blocking=True)
        st["injected" if arrivals is not None else "plain"]()

    # dead code on the MARL path (Environment.py:192-201, 544-545); never called by the driver
    def get_path_loss(self, position_A):
        raise NotImplementedError("get_path_loss is dead code in the reference MARL path (Environment.py:192)")

    def get_shadowing(self, delta_distance, vehicle):
        raise NotImplementedError("get_shadowing is dead code in the reference MARL path (Environment.py:198)")

    def localProcRev(self, b):
        raise NotImplementedError("localProcRev is dead code in the reference MARL path (Environment.py:544)")
