"""`VecEnviron`: E independent RIS-VEC environments resident on one MI355X.

Host-side mirror of the reference's `Environ` class (Environment.py:56) for the hot
path only: same method names, same argument meaning, batched over a leading env
axis and returning torch tensors that live on the GPU.  All arithmetic happens in
the HIP kernels of `csrc/` behind the C ABI of `include/risvec.h`; torch is used
for device memory and streams.  There is no CPU fallback: without a HIP device (or
without the built extension) every compute call raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _native as N
from .params import CHANNEL_MODELS, EnvParams

# module constants of the reference (Environment.py:29-42)
RIS_x, RIS_y, RIS_z = 220, 220, 25
BS_x, BS_y, BS_z = 0, 0, 25
ro = 10 ** -2
lamb = 1
d = 0.5
sigma = 10 ** (-7)
alpha1 = 2.2
alpha2 = 2.5


class ParamAttrs:
    """Gives an env object the reference's flat attribute surface: `env.w_d = 1.0`,
    `env.noise_power` ... are forwarded to `self.params` (an `EnvParams`)."""

    def __getattr__(self, name):
        # only called when normal lookup fails
        params = self.__dict__.get("params")
        if params is not None and name in EnvParams._FIELDS:
            return getattr(params, name)
        raise AttributeError("%s object has no attribute %r" % (type(self).__name__, name))

    def __setattr__(self, name, value):
        params = self.__dict__.get("params")
        if params is not None and name in EnvParams._FIELDS:
            setattr(params, name, value)
        else:
            object.__setattr__(self, name, value)


def _dev_ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class VecEnviron(ParamAttrs):
    """Batched environment.  Constructor mirrors Environment.py:57 plus batching args."""

    def __init__(self, down_lane, up_lane, left_lane, right_lane, width, height, n_veh, M, control_bit,
                 n_envs: int = 1, device: str = "cuda", seed: int = 0, env_offset: int = 0,
                 params: Optional[EnvParams] = None, lazy_theta: bool = False):
        object.__setattr__(self, "params", params if params is not None else EnvParams())
        self.down_lanes = list(down_lane)
        self.up_lanes = list(up_lane)
        self.left_lanes = list(left_lane)
        self.right_lanes = list(right_lane)
        self.width = width
        self.height = height
        self.n_veh = int(n_veh)
        self.M = int(M)
        self.control_bit = int(control_bit)
        self.n_envs = int(n_envs)
        self.env_offset = int(env_offset)
        self.seed = int(seed)
        self.device = N.resolve_device(device)
        if not 1 <= self.n_veh <= N.MAX_VEH:
            raise ValueError("n_veh must be in [1, %d]" % N.MAX_VEH)
        if self.n_envs < 1 or self.M < 1:
            raise ValueError("n_envs and M must be >= 1")
        if not 0 <= self.control_bit <= 6:
            raise ValueError("control_bit must be in [0, 6]")
        # Environment.py:169, 175-179
        self.possible_angles = np.linspace(0, 2 * np.pi, 2 ** self.control_bit, endpoint=False)
        self.distance_B_R = float(np.sqrt((BS_x - RIS_x) ** 2 + (BS_y - RIS_y) ** 2 + (BS_z - RIS_z) ** 2))
        self.angle_B_R = (RIS_x - BS_x) / self.distance_B_R
        m = np.arange(self.M, dtype=np.float64)
        ph = 2 * (np.pi / lamb) * d * self.angle_B_R * m
        self.phase_R = np.cos(ph) + 1j * np.sin(ph)
        self._epoch = 0          # reset counter   (RNG counter for spawn draws)
        self._moves = 0          # renew_positions counter
        self._steps = 0          # step counter    (RNG counter for arrivals)
        self._chan = 0           # 3GPP-gain / random-phase counter
        self._obs_stale = True   # obs[E,V,5] does not reflect the state tensors (no step since the last reset)
        self._t: Dict[str, torch.Tensor] = {}
        self._colsum_valid = False     # c_col matches h_r (set by compute_parms / rebuild_colsum)
        self._steer_valid = False      # h_r is the steering vector compute_parms wrote, z_r its base
        self._ssum_sweeps = 0          # >0: s_sum = sum theta.c of the CURRENT theta, left by that many
                                       # consecutive sweeps (0 = unknown; refreshed every 64 sweeps)
        self._idx_valid = False        # theta_idx = candidate index of every CURRENT theta element (left by a sweep)
        # lazy_theta: between BCD sweeps keep theta BY INDEX (one byte per element, theta_idx): `step(bcd=True)` then
        # neither writes nor reads the complex64 tensor (67 MB out + 67 MB in per step at BASELINE configs[4]); it is
        # materialised when somebody asks for it (`tensors`, any other consumer).  Same step outputs bit for bit.
        self.lazy_theta = bool(lazy_theta)
        self._theta_stale = False      # tensors["theta"] lags behind theta_idx (only ever True with lazy_theta)
        self._tk_ok = None             # does the fused step have a theta-by-index form at this shape? (asked lazily)
        self._cstate: Optional[N.RisVecState] = None
        self._cparams: Optional[N.RisVecParams] = None
        self._cparams_version = -1

    # ------------------------------------------------------------------ device state
    def _lanes(self):
        return dict(up_lanes=self.up_lanes, down_lanes=self.down_lanes, left_lanes=self.left_lanes,
                    right_lanes=self.right_lanes)

    def _ensure_device(self) -> None:
        if self._cstate is not None:
            return
        lib = N.load()       # raises if the extension is not built
        del lib
        self.device = N.resolve_device(self.device)      # a bare "cuda" becomes cuda:<current>
        if self.device.type != "cuda" or not torch.cuda.is_available():
            raise RuntimeError("ris_vec_marl_amd needs a HIP device (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        E, V, M = self.n_envs, self.n_veh, self.M
        dev = self.device
        z = lambda *shape, dt=torch.float32: torch.zeros(*shape, dtype=dt, device=dev)   # noqa: E731
        t = self._t
        t["pos"] = z(E, V, 2, dt=torch.float64)
        t["dir"] = z(E, V, dt=torch.int32)
        t["vel"] = z(E, V)
        t["dist_r"] = z(E, V)
        t["ang_r"] = z(E, V)
        t["pl"] = z(E, V)
        t["h_r"] = z(E, V, M, 2)        # all-zero until compute_parms(), like Environment.py:162
        t["theta"] = z(E, M, 2)         # all-zero start, Environment.py:171
        t["b"] = torch.from_numpy(np.stack([self.phase_R.real, self.phase_R.imag], -1).astype(np.float32)).to(dev)
        # step()'s per-env float32 state and outputs are views into ONE allocation (each view 256-byte aligned): the
        # kernels do not care, and a host that wants them all after a step -- the E = 1 facade, which returns the
        # reference's 7-tuple and 13 last_* scalars as NumPy values -- reads them back with one copy instead of nine
        slab_shapes = [("reward", (E, V)), ("over_power", (E, V)), ("data_buf", (E, V)), ("data_t", (E, V)),
                       ("data_p", (E, V)), ("over_data", (E, V)), ("rate", (E, V)), ("gain", (E, V)), ("mec_q", (E,)),
                       ("metrics", (E, N.METRICS)), ("power_w", (E, 2, V)), ("obs", (E, V, 5))]
        offs, n = {}, 0
        for k, shp in slab_shapes:
            offs[k] = n
            n += (int(np.prod(shp)) + 63) // 64 * 64
        slab = z(n)
        for k, shp in slab_shapes:
            t[k] = slab[offs[k]:offs[k] + int(np.prod(shp))].view(*shp)
        self._out_slab, self._out_offsets = slab, {k: (offs[k], shp) for k, shp in slab_shapes}
        # BCD column sums (sum_v h_r) * b in f64, lane-major slabs of 64 envs: [ceil(E/64), M, 64, 2]
        t["c_col"] = z((E + 63) // 64, M, 64, 2, dt=torch.float64)
        t["s_sum"] = z(E, 2, dt=torch.float64)         # sum_m theta_m c_m left by the last sweep
        t["z_r"] = z(E, V, 2, dt=torch.float64)        # steering base exp(-j pi angle) per vehicle: h_r[e,v,m] = z^m
        # candidate index of every theta element as the last BCD sweep left it: [E, 32 ceil(M/32)] bytes
        t["theta_idx"] = z(E, (M + 31) // 32 * 32, dt=torch.uint8)
        s = N.RisVecState()
        s.abi_version = N.ABI_VERSION
        s.struct_bytes = C.sizeof(N.RisVecState)
        s.n_envs, s.n_veh, s.n_ris, s.control_bit = E, V, M, self.control_bit
        s.env_offset = self.env_offset
        for k in ("pos", "dir", "vel", "dist_r", "ang_r", "pl", "h_r", "theta", "b", "gain", "data_buf",
                  "mec_q", "rate", "data_t", "data_p", "reward", "over_power", "obs", "metrics", "power_w", "c_col",
                  "s_sum", "over_data", "z_r", "theta_idx"):
            setattr(s, k, t[k].data_ptr())
        s.h_d = None
        self._cstate = s
        self._place_stream()

    # how the h_r allocation was chosen (None: not needed / switched off); bench.py reports it
    placement: Optional[dict] = None

    def _place_stream(self) -> None:
        """h_r is THE stream of the fused step.  Once it no longer fits the Infinity Cache, the rate at which the kernels
        stream it depends on WHERE the allocation landed in HBM: two levels ~8 % apart, a property of the allocation (in
        one process the first few GB handed out stream slower than later ones; `tools/placement_probe.py`,
        EXPERIMENTS.md round 3).  So the stream is placed by measurement: up to 10 candidate allocations are timed with
        the fused step kernel itself (on the all-zero state, wiped afterwards) and the fastest is kept; the
        others go back to torch's allocator.  A few milliseconds at construction, at most 25 % of the free memory in
        flight; RISVEC_NO_PLACEMENT=1 switches it off."""
        import os
        t = self._t
        nbytes = t["h_r"].numel() * 4
        if nbytes <= (256 << 20) or os.environ.get("RISVEC_NO_PLACEMENT"):
            return
        free, _ = torch.cuda.mem_get_info(self.device)
        n_max = min(10, 1 + int(0.25 * free // nbytes))
        if n_max < 2:
            return
        lib, cs, pp, stream = N.load(), C.byref(self._cstate), C.byref(self._p()), self._stream()
        E, V = self.n_envs, self.n_veh
        act = torch.zeros(E, 2, V, device=self.device)
        part = torch.full((E, V), -1, dtype=torch.int32, device=self.device)
        ngr = torch.full((E,), V, dtype=torch.int32, device=self.device)
        flags = N.STEP_METRICS | N.STEP_OBS

        def gain_us(h) -> float:                       # the fused step itself on an all-zero state (wiped afterwards)
            self._cstate.h_r = h.data_ptr()
            best = float("inf")
            for i in range(4):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                N.check(lib.risvec_step_fused(cs, pp, act.data_ptr(), part.data_ptr(), ngr.data_ptr(), None, self.seed, i, flags,
                                              stream))
                b.record()
                b.synchronize()
                if i:                                  # the first launch warms the code path
                    best = min(best, a.elapsed_time(b) * 1e3)
            return best

        cands = [(gain_us(t["h_r"]), t["h_r"])]
        while len(cands) < n_max:
            c = torch.zeros_like(cands[0][1])
            cands.append((gain_us(c), c))
            if min(x[0] for x in cands) < 0.96 * max(x[0] for x in cands):
                break                                  # both levels seen: the faster one is known
        us, keep = min(cands, key=lambda x: x[0])
        t["h_r"] = keep
        self._cstate.h_r = keep.data_ptr()
        self.placement = dict(candidates=len(cands), step_kernel_us=[round(x[0], 1) for x in cands], kept_us=round(us, 1))
        self._out_slab.zero_()                         # the timing steps ran on (and dirtied) the all-zero state

    def set_direct_link(self, h_d: Optional[torch.Tensor]) -> None:
        """Optional direct BS link amplitude h_d [E,V] complex64 (the reference has none:
        Environment.py:263-273); None restores the reference behaviour."""
        self._ensure_device()
        if h_d is None:
            self._t.pop("h_d", None)
            self._cstate.h_d = None
            return
        h = torch.view_as_real(h_d.to(self.device, torch.complex64)).contiguous()
        if tuple(h.shape) != (self.n_envs, self.n_veh, 2):
            raise ValueError("h_d must have shape [n_envs, n_veh]")
        self._t["h_d"] = h
        self._cstate.h_d = h.data_ptr()

    def _p(self) -> N.RisVecParams:
        if self._cparams is None or self._cparams_version != self.params.version:
            self._cparams = self.params.to_c(self._lanes(), self.width, self.height)
            self._cparams_version = self.params.version
        return self._cparams

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def _arg(self, x, dtype, shape, name) -> Optional[torch.Tensor]:
        if x is None:
            return None
        tt = torch.as_tensor(x)
        if tuple(tt.shape) != tuple(shape):
            raise ValueError("%s must have shape %s, got %s" % (name, tuple(shape), tuple(tt.shape)))
        return tt.to(device=self.device, dtype=dtype).contiguous()

    def _bound(self, x, dtype, shape, name) -> Optional[torch.Tensor]:
        """An input of a pre-marshalled launcher: used as is or refused, never copied."""
        if x is None:
            return None
        if (not isinstance(x, torch.Tensor) or x.dtype != dtype or x.device != self.device or not x.is_contiguous()
                or tuple(x.shape) != tuple(shape)):
            raise ValueError("%s must be a contiguous %s tensor of shape %s on %s (bound launchers read their inputs "
                             "in place; convert it once before binding)" % (name, dtype, tuple(shape), self.device))
        return x

    # ------------------------------------------------------------------ tensors (views)
    @property
    def tensors(self) -> Dict[str, torch.Tensor]:
        self._ensure_device()
        self._sync_theta()
        return self._t

    def _sync_theta(self) -> None:
        """Materialise tensors["theta"] from the candidate indices if the last sweeps kept theta by index."""
        if self._theta_stale:
            N.check(N.load().risvec_theta_from_index(C.byref(self._cstate), self._stream()))
            self._theta_stale = False

    def _by_index(self, fused: bool, steer: bool) -> bool:
        """Can this fused step read theta as candidate indices?  (lazy_theta, indices current, a shape with that form)"""
        if not (self.lazy_theta and fused and not steer and self._idx_valid and self.control_bit == 3):
            return False
        if self._tk_ok is None:                  # asked once: the answer depends on the shape only
            self._tk_ok = bool(N.load().risvec_theta_by_index_supported(self.n_veh, self.M))
        return self._tk_ok

    def __getattr__(self, name):
        t = self.__dict__.get("_t")
        alias = _TENSOR_ALIASES.get(name)
        if alias is not None and t is not None:
            self._ensure_device()
            return self._t[alias]
        return ParamAttrs.__getattr__(self, name)

    @property
    def elements_phase_shift_complex(self) -> torch.Tensor:
        return torch.view_as_complex(self.tensors["theta"])

    @property
    def phases_R_i(self) -> torch.Tensor:
        return torch.view_as_complex(self.tensors["h_r"])

    # ------------------------------------------------------------------ reference methods
    def make_new_game(self, spawn_ints=None, buf0=None) -> None:
        """Environment.py:733-737 (+381-410).  Optional injected draws: spawn_ints
        [E,V,3] int32 (aux, coord, velocity), buf0 [E] int32."""
        self._ensure_device()
        E, V = self.n_envs, self.n_veh
        si = self._arg(spawn_ints, torch.int32, (E, V, 3), "spawn_ints")
        b0 = self._arg(buf0, torch.int32, (E,), "buf0")
        self._epoch += 1
        N.check(N.load().risvec_reset(C.byref(self._cstate), C.byref(self._p()), _dev_ptr(si), _dev_ptr(b0),
                                      self.seed, self._epoch, self._stream()))
        self._obs_stale = True         # DataBuf changed under the observation the last step wrote

    def renew_positions(self, u_turn=None, return_n_used: bool = False):
        """Environment.py:412-542.  u_turn [E,V,8] float32 injected uniform draws."""
        self._ensure_device()
        E, V = self.n_envs, self.n_veh
        u = self._arg(u_turn, torch.float32, (E, V, 8), "u_turn")
        nu = torch.zeros(E, V, dtype=torch.int32, device=self.device) if return_n_used else None
        self._moves += 1
        N.check(N.load().risvec_mobility(C.byref(self._cstate), C.byref(self._p()), _dev_ptr(u), _dev_ptr(nu),
                                         self.seed, self._moves, self._stream()))
        return nu

    def compute_parms(self) -> None:
        """Environment.py:241-253: pos -> distances_R_i, angles_R_i, phases_R_i (+ path-loss factor)."""
        self._ensure_device()
        N.check(N.load().risvec_geometry(C.byref(self._cstate), C.byref(self._p()), self._stream()))
        self._colsum_valid = True
        self._steer_valid = True       # h_r[e,v,m] = z_r[e,v]^m from here on
        self._ssum_sweeps = 0          # c changed: the cached sum is stale (the candidate indices are not)

    def rebuild_colsum(self) -> None:
        """Recompute the BCD cache c_col[e,m] = (sum_v h_r[e,v,m]) b[m] (float64).  compute_parms()
        does it already; call this after writing `tensors["h_r"]` directly (which also ends the
        validity of the steering form of the fused step, `steer=True`)."""
        self._ensure_device()
        N.check(N.load().risvec_colsum(C.byref(self._cstate), self._stream()))
        self._colsum_valid = True
        self._steer_valid = False
        self._ssum_sweeps = 0

    def colsum_rows(self) -> torch.Tensor:
        """The BCD cache as [E, M] complex128 (a copy; the device layout is lane-major slabs)."""
        c = torch.view_as_complex(self.tensors["c_col"])            # [S, M, 64]
        return c.permute(0, 2, 1).reshape(-1, self.M)[: self.n_envs].clone()

    def invalidate_colsum(self) -> None:
        """Tell the env that h_r or theta was modified behind its back (a direct write to
        `tensors[...]`): the next BCD rebuilds c_col, re-sums theta.c and re-derives the candidate indices."""
        self._colsum_valid = False
        self._theta_changed()

    def invalidate_theta(self) -> None:
        """Tell the env that `tensors["theta"]` was written directly: the next BCD sweep re-sums theta.c and
        re-derives the candidate indices (the phase setters and the sweeps themselves keep track on their own)."""
        self._theta_changed()

    def _theta_changed(self) -> None:
        """theta was written by something other than a BCD sweep: its cached sum and candidate indices are stale."""
        self._ssum_sweeps = 0
        self._idx_valid = False
        self._theta_stale = False      # whoever wrote theta made the tensor the truth again

    def _bcd_flags(self, reuse_colsum: Optional[bool], step: bool) -> int:
        reuse_c = self._colsum_valid if reuse_colsum is None else bool(reuse_colsum)
        reuse_s = reuse_c and 0 < self._ssum_sweeps < 64
        reuse_i = self._idx_valid and self.control_bit == 3
        if step:
            return ((N.STEP_REUSE_COLSUM if reuse_c else 0) | (N.STEP_REUSE_SSUM if reuse_s else 0)
                    | (N.STEP_REUSE_IDX if reuse_i else 0))
        return ((N.BCD_REUSE_COLSUM if reuse_c else 0) | (N.BCD_REUSE_SSUM if reuse_s else 0)
                | (N.BCD_REUSE_IDX if reuse_i else 0))

    def _bcd_done(self, flags: int, step: bool) -> None:
        reused_s = bool(flags & (N.STEP_REUSE_SSUM if step else N.BCD_REUSE_SSUM))
        self._colsum_valid = True
        self._ssum_sweeps = self._ssum_sweeps + 1 if reused_s else 1
        self._idx_valid = self.control_bit == 3        # every 2^b = 8 sweep leaves the indices of what it stored

    def optimize_phase_shift(self, return_idx: bool = False, reuse_colsum: Optional[bool] = None):
        """Environment.py:208-220 (one BCD sweep, objective of :222-231).  The column sums the
        sweep needs are pure geometry; they are reused when known current (after
        compute_parms()/rebuild_colsum()), otherwise rebuilt first.  reuse_colsum overrides."""
        self._ensure_device()
        idx = torch.zeros(self.n_envs, self.M, dtype=torch.int32, device=self.device) if return_idx else None
        flags = self._bcd_flags(reuse_colsum, step=False)
        lazy = self.lazy_theta and bool(flags & N.BCD_REUSE_IDX)
        if lazy:
            flags |= N.BCD_NO_THETA        # the indices are the state; theta follows on demand
        N.check(N.load().risvec_bcd(C.byref(self._cstate), C.byref(self._p()), _dev_ptr(idx), flags, self._stream()))
        self._bcd_done(flags, step=False)
        self._theta_stale = lazy
        return idx

    def update_channel_gains(self, u_los=None, z_shadow=None, small=None) -> None:
        """Environment.py:255-327, dispatching on `channel_model` like the reference
        (unknown keywords fall back to a 0 dB path loss, :315-317)."""
        self._ensure_device()
        model = str(self.params.channel_model)
        if model == "free":
            self._sync_theta()
            N.check(N.load().risvec_gain(C.byref(self._cstate), C.byref(self._p()), self._stream()))
            return
        E, V = self.n_envs, self.n_veh
        ul = self._arg(u_los, torch.float32, (E, V), "u_los")
        zs = self._arg(z_shadow, torch.float32, (E, V), "z_shadow")
        sm = self._arg(small, torch.float32, (E, V), "small")
        self._chan += 1
        N.check(N.load().risvec_gain_3gpp(C.byref(self._cstate), C.byref(self._p()),
                                          CHANNEL_MODELS.get(model, N.CH_OTHER), _dev_ptr(ul), _dev_ptr(zs),
                                          _dev_ptr(sm), self.seed, self._chan, self._stream()))

    def get_channel_gains(self) -> torch.Tensor:
        """Environment.py:374-376."""
        return self.tensors["gain"]

    def get_next_phase(self, action_phase) -> None:
        """Environment.py:233-239: theta = exp(j * angle), angle [E,M]."""
        self._ensure_device()
        a = self._arg(action_phase, torch.float32, (self.n_envs, self.M), "action_phase")
        N.check(N.load().risvec_set_phase(C.byref(self._cstate), _dev_ptr(a), self._stream()))
        self._theta_changed()

    def Random_phase(self, idx=None) -> None:
        """Environment.py:203-206; idx [E,M] int32 indices into possible_angles (optional)."""
        self._ensure_device()
        i = self._arg(idx, torch.int32, (self.n_envs, self.M), "idx")
        self._chan += 1
        N.check(N.load().risvec_random_phase(C.byref(self._cstate), _dev_ptr(i), self.seed, self._chan,
                                             self._stream()))
        self._theta_changed()

    def data_rate(self, p_off, partner, n_groups) -> torch.Tensor:
        """Environment.py:331-372 on the cached gains: p_off [E,V] offload power in W -> rate [E,V]."""
        self._ensure_device()
        E, V = self.n_envs, self.n_veh
        pw = self._arg(p_off, torch.float32, (E, V), "p_off")
        pt = self._arg(partner, torch.int32, (E, V), "partner")
        ng = self._arg(n_groups, torch.int32, (E,), "n_groups")
        out = torch.empty(E, V, dtype=torch.float32, device=self.device)
        N.check(N.load().risvec_data_rate(C.byref(self._cstate), C.byref(self._p()), _dev_ptr(pw), _dev_ptr(pt),
                                          _dev_ptr(ng), _dev_ptr(out), self._stream()))
        return out

    def _steer_flag(self, steer: bool, fused: bool) -> int:
        if not steer:
            return 0
        if not fused:
            raise ValueError("steer=True is a form of the fused gain+step kernel: pass fused=True")
        if not self._steer_valid:
            raise ValueError("steer=True needs h_r to be the steering vectors compute_parms() wrote "
                             "(call compute_parms(); h_r written by hand has no steering base)")
        return N.STEP_STEER

    def step(self, action_power, partner, n_groups, arrivals=None, fused: bool = False, bcd: bool = False,
             metrics: bool = True, power_w: bool = True, obs: bool = True, policy_action: bool = False,
             steer: bool = False) -> Tuple[torch.Tensor, ...]:
        """Environment.py:547-731 for every env.

        action_power [E,2,V] float32 (or the policy output [E,V,2] with policy_action=True,
        marl_train_bcd.py:1601-1608); partner [E,V] int32 / n_groups [E] int32 encode
        `noma_groups` (see `compat.encode_noma_groups`); arrivals [E,V] int32 = injected
        Poisson draws (None: in-kernel Philox).  fused=True recomputes the RIS cascaded
        gains in the same launch (the north-star kernel); bcd=True additionally runs a BCD
        sweep first; steer=True (with fused) uses the fact that compute_parms made every h_r row a
        geometric sequence z^m and evaluates the cascade by Horner in float64 from the 16-byte base
        instead of reading the 8M-byte row (same results to ~1e-7; ~4x fewer bytes per step).
        Returns the reference's 7-tuple, batched:
        (per_user_reward [E,V], global_reward [E], DataBuf, data_t, data_p, over_power, over_data);
        the tensors are owned by the env and overwritten by the next step."""
        self._ensure_device()
        E, V = self.n_envs, self.n_veh
        a = self._arg(action_power, torch.float32, (E, V, 2) if policy_action else (E, 2, V), "action_power")
        pt = self._arg(partner, torch.int32, (E, V), "partner")
        ng = self._arg(n_groups, torch.int32, (E,), "n_groups")
        ar = self._arg(arrivals, torch.int32, (E, V), "arrivals")
        flags = ((N.STEP_METRICS if metrics else 0) | (N.STEP_POWER_W if power_w else 0)
                 | (N.STEP_OBS if obs else 0) | (N.STEP_POLICY_ACTION if policy_action else 0)
                 | (self._bcd_flags(None, step=True) if bcd else 0) | self._steer_flag(steer, fused or bcd))
        flags = self._theta_mode(flags, fused or bcd, bcd, steer)
        lib = N.load()
        fn = lib.risvec_step_fused_bcd if bcd else (lib.risvec_step_fused if fused else lib.risvec_step)
        N.check(fn(C.byref(self._cstate), C.byref(self._p()), _dev_ptr(a), _dev_ptr(pt), _dev_ptr(ng),
                   _dev_ptr(ar), self.seed, self._steps, flags, self._stream()))
        if bcd:
            self._bcd_done(flags, step=True)
            self._theta_stale = bool(flags & N.STEP_THETA_BY_INDEX)
        self._steps += 1
        self._obs_stale = not obs
        t = self._t
        return (t["reward"], t["metrics"][:, 0], t["data_buf"], t["data_t"], t["data_p"], t["over_power"],
                t["over_data"])

    def _theta_mode(self, flags: int, fused: bool, bcd: bool, steer: bool) -> int:
        """Decide how a step gets at theta: by index (flag added) where lazy_theta allows, else make sure the complex64
        tensor is current for the kernels that read it."""
        if self._by_index(fused, steer) and (not bcd or bool(flags & N.STEP_REUSE_IDX)):
            return flags | N.STEP_THETA_BY_INDEX
        if fused:
            self._sync_theta()
        return flags

    def step_many(self, actions, partner, n_groups, arrivals=None, metrics: bool = True, power_w: bool = False,
                  obs: bool = True, policy_action: bool = False, record: Sequence[str] = ("reward", "obs", "metrics"),
                  out: Optional[Dict[str, torch.Tensor]] = None, fused: bool = True) -> Dict[str, torch.Tensor]:
        """T consecutive fused `step()` calls in ONE launch (`risvec_step_fused_multi`): the driver's step loop
        marl_train_bcd.py:1304-1611 between two channel refreshes with the NOMA groups frozen, as they are
        inside an episode.  actions [T,E,2,V] float32 (or [T,E,V,2] with policy_action=True); partner / n_groups as
        for `step()` (the same for every step); arrivals [T,E,V] int32 injected draws or None (Philox with the
        counters T single calls would use).  The env's tensors end up exactly as after the last of T single
        `step(..., fused=True)` calls, bit for bit; `record` names the per-step records to keep
        ("reward" [T,E,V], "obs" [T,E,V,5], "metrics" [T,E,16]) -- returned as a dict, written into `out`'s
        tensors when given.  h_r / theta cannot change inside the launch, so the gains are computed once and each
        env's queues stay in registers: this is the launch to use when a batched step is shorter than a kernel
        launch (small E).  fused=False (`risvec_step_multi`) is the same on the CACHED gains -- T `step(...,
        fused=False)` calls, the reference driver's own cadence (gains only every 100 steps), any shape, h_r / theta
        not read at all."""
        self._ensure_device()
        if fused:
            self._sync_theta()
        E, V = self.n_envs, self.n_veh
        a = torch.as_tensor(actions)
        if a.dim() != 4 or tuple(a.shape[1:]) != ((E, V, 2) if policy_action else (E, 2, V)):
            raise ValueError("actions must have shape [T, %d, %s]" % (E, "%d, 2" % V if policy_action else "2, %d" % V))
        T = int(a.shape[0])
        if T < 1:
            raise ValueError("actions holds no step")
        a = a.to(device=self.device, dtype=torch.float32).contiguous()
        pt = self._arg(partner, torch.int32, (E, V), "partner")
        ng = self._arg(n_groups, torch.int32, (E,), "n_groups")
        ar = self._arg(arrivals, torch.int32, (T, E, V), "arrivals")
        shapes = {"reward": (T, E, V), "obs": (T, E, V, 5), "metrics": (T, E, N.METRICS)}
        rec: Dict[str, torch.Tensor] = {}
        for k in record:
            if k not in shapes:
                raise ValueError("record: unknown trajectory record %r (choose from %s)" % (k, sorted(shapes)))
            if k == "obs" and not obs:
                raise ValueError("record 'obs' needs obs=True")
            if out is not None and k in out:
                rec[k] = self._bound(out[k], torch.float32, shapes[k], "out[%r]" % k)
            else:
                rec[k] = torch.empty(shapes[k], dtype=torch.float32, device=self.device)
        tj = N.RisVecTraj(_dev_ptr(rec.get("reward")), _dev_ptr(rec.get("obs")), _dev_ptr(rec.get("metrics")))
        flags = ((N.STEP_METRICS if metrics else 0) | (N.STEP_POWER_W if power_w else 0)
                 | (N.STEP_OBS if obs else 0) | (N.STEP_POLICY_ACTION if policy_action else 0))
        fn = N.load().risvec_step_fused_multi if fused else N.load().risvec_step_multi
        N.check(fn(C.byref(self._cstate), C.byref(self._p()), T, _dev_ptr(a), _dev_ptr(pt), _dev_ptr(ng), _dev_ptr(ar),
                   self.seed, self._steps, flags, C.byref(tj), self._stream()))
        self._steps += T
        self._obs_stale = not obs
        return rec

    def bind_step_many(self, actions: torch.Tensor, partner: torch.Tensor, n_groups: torch.Tensor,
                       arrivals: Optional[torch.Tensor] = None, metrics: bool = True, power_w: bool = False,
                       obs: bool = True, policy_action: bool = False,
                       out: Optional[Dict[str, torch.Tensor]] = None, fused: bool = True):
        """`step_many` validated and marshalled once: returns a zero-argument launcher that advances the env by
        T steps per call, reading `actions` [T,E,...] (and `arrivals`) in place and writing the per-step records
        into `out`'s tensors ("reward" [T,E,V], "obs" [T,E,V,5], "metrics" [T,E,16]; any subset)."""
        self._ensure_device()
        E, V = self.n_envs, self.n_veh
        if not isinstance(actions, torch.Tensor) or actions.dim() != 4:
            raise ValueError("actions must be a [T, E, ...] device tensor")
        T = int(actions.shape[0])
        a = self._bound(actions, torch.float32, (T, E, V, 2) if policy_action else (T, E, 2, V), "actions")
        pt = self._bound(partner, torch.int32, (E, V), "partner")
        ng = self._bound(n_groups, torch.int32, (E,), "n_groups")
        ar = self._bound(arrivals, torch.int32, (T, E, V), "arrivals")
        shapes = {"reward": (T, E, V), "obs": (T, E, V, 5), "metrics": (T, E, N.METRICS)}
        rec = {k: self._bound(t, torch.float32, shapes[k], "out[%r]" % k) for k, t in (out or {}).items()}
        tj = N.RisVecTraj(_dev_ptr(rec.get("reward")), _dev_ptr(rec.get("obs")), _dev_ptr(rec.get("metrics")))
        flags = ((N.STEP_METRICS if metrics else 0) | (N.STEP_POWER_W if power_w else 0)
                 | (N.STEP_OBS if obs else 0) | (N.STEP_POLICY_ACTION if policy_action else 0))
        fn = N.load().risvec_step_fused_multi if fused else N.load().risvec_step_multi
        cs, seed, stream = C.byref(self._cstate), C.c_uint64(self.seed), self._stream()
        pa, pp, pn, par, ptj = _dev_ptr(a), _dev_ptr(pt), _dev_ptr(ng), _dev_ptr(ar), C.byref(tj)

        def launch() -> None:
            if fused:
                self._sync_theta()
            rc = fn(cs, C.byref(self._p()), T, pa, pp, pn, par, seed, self._steps, flags, ptj, stream)
            if rc:
                N.check(rc)
            self._steps += T
            self._obs_stale = not obs

        launch.inputs = (a, pt, ng, ar, rec, tj)
        launch.n_steps = T
        return launch

    def sarl_step(self, action_power, action_phase=None, arrivals=None, sarl_params=None, obs: bool = True
                  ) -> Tuple[torch.Tensor, ...]:
        """The single-agent variant's step (Simulation-SARL/Environment.py:321-359) for every env:
        action_power [E,2,V] float32 used as given, action_phase [E,M] radians (None keeps the
        current theta), arrivals [E,V] int32 injected Poisson draws (None: Philox).  Returns the
        reference's 6-tuple, batched: (Reward [E], DataBuf, data_t, data_p, over_power, over_data)."""
        from .sarl import SarlParams
        self._ensure_device()
        E, V, M = self.n_envs, self.n_veh, self.M
        a = self._arg(action_power, torch.float32, (E, 2, V), "action_power")
        ph = self._arg(action_phase, torch.float32, (E, M), "action_phase")
        ar = self._arg(arrivals, torch.int32, (E, V), "arrivals")
        sp = (sarl_params or SarlParams()).to_c()
        self._sync_theta()
        N.check(N.load().risvec_sarl_step(C.byref(self._cstate), C.byref(sp), _dev_ptr(a), _dev_ptr(ph),
                                          _dev_ptr(ar), self.seed, self._steps, N.STEP_OBS if obs else 0,
                                          self._stream()))
        if ph is not None:
            self._theta_changed()
        self._steps += 1
        self._obs_stale = False        # sarl_observe assembles its own observation from the state tensors
        t = self._t
        return (t["metrics"][:, 0], t["data_buf"], t["data_t"], t["data_p"], t["over_power"], t["over_data"])

    def bind_sarl_step(self, action_power: torch.Tensor, action_phase: Optional[torch.Tensor] = None,
                       arrivals: Optional[torch.Tensor] = None, sarl_params=None, obs: bool = True):
        """`sarl_step` validated and marshalled once: returns a zero-argument launcher that reads the SAME device
        tensors on every call (update them in place between calls).  The parameters are those of `sarl_params` at
        bind time."""
        from .sarl import SarlParams
        self._ensure_device()
        E, V, M = self.n_envs, self.n_veh, self.M
        a = self._bound(action_power, torch.float32, (E, 2, V), "action_power")
        ph = self._bound(action_phase, torch.float32, (E, M), "action_phase")
        ar = self._bound(arrivals, torch.int32, (E, V), "arrivals")
        sp = (sarl_params or SarlParams()).to_c()
        fn, cs, psp, seed, stream = N.load().risvec_sarl_step, C.byref(self._cstate), C.byref(sp), C.c_uint64(self.seed), self._stream()
        pa, pph, par, flags = _dev_ptr(a), _dev_ptr(ph), _dev_ptr(ar), N.STEP_OBS if obs else 0

        def launch() -> None:
            self._sync_theta()
            rc = fn(cs, psp, pa, pph, par, seed, self._steps, flags, stream)
            if rc:
                N.check(rc)
            if ph is not None:
                self._theta_changed()
            self._steps += 1
            self._obs_stale = False

        launch.inputs = (a, ph, ar, sp)
        return launch

    def bind_step(self, action_power, partner, n_groups, arrivals=None, fused: bool = False, bcd: bool = False,
                  metrics: bool = True, power_w: bool = True, obs: bool = True, policy_action: bool = False,
                  steer: bool = False):
        """Validate and marshal a `step()` call ONCE and return a zero-argument callable that
        launches one step per call on the stream current at bind time, reading the SAME input
        tensors each time (update them in place between calls).  Cuts the per-step host cost
        from ~10 us of Python argument handling to one ctypes call, which matters when a
        batched step is only a few microseconds of GPU time (small E)."""
        self._ensure_device()
        E, V = self.n_envs, self.n_veh
        # the launcher reads these tensors IN PLACE on every call, so they must already be what the
        # kernel reads: a silent .to()/.contiguous() copy would detach the caller's later updates
        a = self._bound(action_power, torch.float32, (E, V, 2) if policy_action else (E, 2, V), "action_power")
        pt = self._bound(partner, torch.int32, (E, V), "partner")
        ng = self._bound(n_groups, torch.int32, (E,), "n_groups")
        ar = self._bound(arrivals, torch.int32, (E, V), "arrivals")
        base_flags = ((N.STEP_METRICS if metrics else 0) | (N.STEP_POWER_W if power_w else 0)
                      | (N.STEP_OBS if obs else 0) | (N.STEP_POLICY_ACTION if policy_action else 0)
                      | self._steer_flag(steer, fused or bcd))
        lib = N.load()
        fn = lib.risvec_step_fused_bcd if bcd else (lib.risvec_step_fused if fused else lib.risvec_step)
        cs, seed, stream = C.byref(self._cstate), C.c_uint64(self.seed), self._stream()
        pa, pp, pn, par = _dev_ptr(a), _dev_ptr(pt), _dev_ptr(ng), _dev_ptr(ar)
        keep = (a, pt, ng, ar)           # the closure owns the marshalled tensors

        def launch() -> None:
            flags = base_flags | (self._bcd_flags(None, step=True) if bcd else 0)
            if self.lazy_theta or self._theta_stale:
                flags = self._theta_mode(flags, fused or bcd, bcd, steer)
            rc = fn(cs, C.byref(self._p()), pa, pp, pn, par, seed, self._steps, flags, stream)
            if rc:
                N.check(rc)
            if bcd:
                self._bcd_done(flags, step=True)
                self._theta_stale = bool(flags & N.STEP_THETA_BY_INDEX)
            self._steps += 1
            self._obs_stale = not obs

        launch.inputs = keep
        return launch

    def bind_step_store(self, replay, power_raw: torch.Tensor, partner: torch.Tensor, n_groups: torch.Tensor,
                        probs: torch.Tensor, mask: Optional[torch.Tensor] = None, arrivals: Optional[torch.Tensor] = None,
                        fused: bool = True, metrics: bool = True, power_w: bool = False):
        """The rollout step with the transition store fused in (`risvec_step_ring`; marl_train_bcd.py:1601-1611,
        1776-1799): ONE launch runs `step()` on the raw policy output `power_raw` [E,V,2] and appends this step's E
        transitions to `replay` (a `VecReplayBuffer` with input_shape 5, n_actions V+2, n_agents V) -- state = the
        observation the env holds when the launch starts, action row = [probs_i with zero diagonal | raw power_i],
        rewards and the new observation straight from the step's registers.  Ring contents and env tensors are those
        of `bind_step(policy_action=True)` followed by `replay.bind_store(..., policy_out=(power_raw, probs))`, bit
        for bit.  Returns `launch(done=False, use_mask=True)`; all tensors are read in place on every call.
        fused=True needs a shape with a software-pipelined kernel ((8,64), (8,36), (8,40), (4,16), (16,64), (16,256));
        fused=False steps on the cached gains, any M."""
        self._ensure_device()
        E, V = self.n_envs, self.n_veh
        a = self._bound(power_raw, torch.float32, (E, V, 2), "power_raw")
        pt = self._bound(partner, torch.int32, (E, V), "partner")
        ng = self._bound(n_groups, torch.int32, (E,), "n_groups")
        pr = self._bound(probs, torch.float32, (E, V, V), "probs")
        mk = self._bound(mask, torch.uint8, (E, V, V), "mask")
        ar = self._bound(arrivals, torch.int32, (E, V), "arrivals")
        if replay.device != self.device or replay.n_agents != V or replay.input_shape != 5 or replay.n_actions != V + 2:
            raise ValueError("bind_step_store: the replay buffer must live on %s with n_agents=%d, input_shape=5, n_actions=%d"
                             % (self.device, V, V + 2))
        flags = ((N.STEP_METRICS if metrics else 0) | (N.STEP_POWER_W if power_w else 0) | N.STEP_OBS | N.STEP_POLICY_ACTION)
        ring = N.RisVecStepRing()
        ring.rb = replay._c
        ring.probs, ring.mask = pr.data_ptr(), None
        fn, cs, seed, stream = N.load().risvec_step_ring, C.byref(self._cstate), C.c_uint64(self.seed), self._stream()
        pa, pp, pn, par, pmask = _dev_ptr(a), _dev_ptr(pt), _dev_ptr(ng), _dev_ptr(ar), _dev_ptr(mk)
        fz = 1 if fused else 0

        def launch(done: bool = False, use_mask: bool = True) -> None:
            if self._obs_stale:
                self.observe()                   # the ring's `state` is the observation tensor as the kernel finds it
            if fused:
                self._sync_theta()
            ring.mem_cntr, ring.done = replay.mem_cntr, 1 if done else 0
            ring.mask = pmask if use_mask else None
            rc = fn(cs, C.byref(self._p()), C.byref(ring), pa, pp, pn, par, seed, self._steps, flags, fz, stream)
            if rc:
                N.check(rc)
            replay.mem_cntr += E
            self._steps += 1
            self._obs_stale = False

        launch.inputs = (a, pt, ng, pr, mk, ar, ring, replay)
        return launch

    # ------------------------------------------------------------------ driver-side helpers
    def observe(self) -> torch.Tensor:
        """marl_train_bcd.py:819-827 for all agents: [E,V,5].  After a step the kernel has
        already written it; before the first step it is assembled from the state."""
        t = self.tensors
        if self._obs_stale:
            # what marl_get_state reads at this point: the CURRENT attributes (a reset replaces DataBuf
            # and leaves data_t / data_p / vehicle_rate of the last step alone, Environment.py:733-737)
            o = t["obs"]
            o[..., 0] = t["data_buf"] / 10
            o[..., 1] = t["data_t"] / 10
            o[..., 2] = t["data_p"] / 10
            o[..., 3] = 0
            o[..., 4] = t["rate"] / 20
            self._obs_stale = False
        return t["obs"]

    def metrics_dict(self) -> Dict[str, torch.Tensor]:
        """The 13 `last_*` scalars + global_reward, each [E] (Environment.py:612-677, 706-711)."""
        m = self.tensors["metrics"]
        return {name: m[:, i] for i, name in enumerate(N.METRIC_NAMES)}

    def begin_episode(self, i_episode: int, env_refresh_every: int = 5) -> bool:
        """Call cadence of marl_train_bcd.py:1268-1271."""
        if i_episode % max(1, int(env_refresh_every)) == 0:
            self.renew_positions()
            self.compute_parms()
            return True
        return False

    def begin_step(self, i_step: int, ris_every: int = 100) -> bool:
        """Call cadence of marl_train_bcd.py:1307-1309 (K_STEPS_FOR_RIS_OPTIMIZATION = 100)."""
        if i_step % max(1, int(ris_every)) == 0:
            self.optimize_phase_shift()
            self.update_channel_gains()
            return True
        return False

    # ------------------------------------------------------------------ checkpoint
    _STATE_KEYS = ("pos", "dir", "vel", "dist_r", "ang_r", "pl", "h_r", "z_r", "theta", "gain", "data_buf", "mec_q",
                   "rate", "data_t", "data_p", "reward", "over_power", "over_data", "obs", "metrics", "power_w")

    def state_dict(self) -> Dict[str, object]:
        t = self.tensors
        sd: Dict[str, object] = {k: t[k].detach().cpu().clone() for k in self._STATE_KEYS}
        sd["counters"] = dict(epoch=self._epoch, moves=self._moves, steps=self._steps, chan=self._chan,
                              seed=self.seed, env_offset=self.env_offset, steer_valid=self._steer_valid,
                              obs_stale=self._obs_stale)
        return sd

    def load_state_dict(self, sd: Dict[str, object]) -> None:
        """Restore a `state_dict()`.  The Philox streams are keyed by (seed, global env id), so a
        checkpoint only resumes bit-identically in an env built with the SAME seed and env_offset:
        a mismatch raises instead of silently diverging."""
        c = sd["counters"]
        for key, mine in (("seed", self.seed), ("env_offset", self.env_offset)):
            if key in c and int(c[key]) != int(mine):
                raise ValueError("load_state_dict: checkpoint was taken with %s=%d, this env has %s=%d (the RNG "
                                 "streams are keyed by it; build the env with the checkpoint's value)"
                                 % (key, int(c[key]), key, int(mine)))
        t = self.tensors
        for k in self._STATE_KEYS:
            if k in sd:                # power_w joined the checkpoint in round 2
                t[k].copy_(sd[k])
        self._colsum_valid = False
        self._theta_changed()
        self._epoch, self._moves, self._steps, self._chan = c["epoch"], c["moves"], c["steps"], c["chan"]
        self._steer_valid = bool(c.get("steer_valid", False))      # z_r travels with h_r
        self._obs_stale = bool(c.get("obs_stale", self._steps == 0))


# reference attribute name -> tensor key
_TENSOR_ALIASES = {
    "DataBuf": "data_buf", "data_t": "data_t", "data_p": "data_p", "over_data": "over_data",
    "vehicle_rate": "rate", "channel_gains": "gain", "distances_R_i": "dist_r", "angles_R_i": "ang_r",
    "mec_queue_cycles": "mec_q", "last_power_W": "power_w", "positions": "pos", "directions": "dir",
    "velocities": "vel",
}
