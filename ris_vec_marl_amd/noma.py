"""f2: the NOMA grouping stage of the reference driver -- what `marl_train_bcd.py`
(TRAIN below) computes for its single env right before every `env.step()` (TRAIN:1317-1343,
1401-1562) -- for all E envs of a `VecEnviron`, on the GPU (`csrc/k_noma.hip`, one wavefront per
env).  The reference has no function boundary here (it is inline script code); `NomaGrouper`
packages that code's episode-scoped variables and its three moments:

    grouper = NomaGrouper(env)                   # config: driver `Config` defaults, or .apply_yaml(y)
    grouper.begin_episode(i_episode)             # TRAIN:1282-1300
    for i_step in range(n_steps):
        if i_step % K_STEPS_FOR_RIS_OPTIMIZATION == 0:
            env.optimize_phase_shift(); env.update_channel_gains()
            mask = grouper.refresh_mask()        # TRAIN:1319-1343 -> [E,N,N] uint8 (also the policy's action mask)
        partner, n_groups = grouper.group(p_off01, i_step)     # TRAIN:1401-1562
        env.step(action, partner, n_groups)      # the global reward it leaves is picked up by the next group()

There is no CPU path: every method launches HIP kernels through the C ABI (`risvec_noma_*`).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Any, Mapping, Optional, Tuple

import torch

from . import _native as N


def anneal_topk(i_ep: int, n_agents: int, k_start: int, k_end: int, T_ep: int) -> int:
    """TRAIN:128-132: linear curriculum of the per-row Top-K (host scalar)."""
    i = max(0, min(int(i_ep), int(T_ep)))
    k = round(k_end + (k_start - k_end) * (1.0 - i / max(1, T_ep)))
    return int(min(max(k, 1), n_agents - 1))


class NomaConfig:
    """The grouping-related attributes of the driver's `Config` (TRAIN:435-441, 489-503) with the
    `getattr` fallbacks the pairing code uses (TRAIN:1404-1417, 1450, 1481-1483, 1499)."""

    def __init__(self, n_veh: int):
        self.n_veh = int(n_veh)
        self.use_mwm_primary = True
        self.mwm_allow_singles = True
        self.mwm_accept_quantile = 0.10
        self.mwm_backoff_rounds = 5
        self.mwm_accept_q_step = 0.05
        self.min_pair_target = max(1, self.n_veh // 4)
        self.completion_min_quantile = 0.30
        self.pairing_threshold_quantile = 0.5
        self.mask_enable = True
        self.mask_topk_start = self.n_veh - 1
        self.mask_topk_end = max(4, self.n_veh // 2)
        self.mask_tau_q_start = 0.2
        self.mask_tau_q_end = 0.4
        self.mask_warmup_episodes = 200
        self.qos_enable = False
        self.qos_R_min_bpsHz = 0.0
        self.qos_soft_penalty_dbscore = 6.0
        self.score_w_delta_db = 1.0
        self.score_w_history = 0.3
        self.abs_gain_min_db = -math.inf
        self.pair_hist_decay = 0.97
        self.relax_q_step = 0.02
        self.relax_topk_step = 1
        self.relax_tau_factor_per_round = 0.95
        self.tau_back_floor_db = 3.0
        self.freeze_group_in_episode = True
        self.freeze_recalc_every = 0
        self.freeze_unstick_prob = 0.0
        self.freeze_reward_drop_ratio = 0.05

    def apply_yaml(self, y: Optional[Mapping[str, Any]]) -> "NomaConfig":
        """The same keys, from the same places, as TRAIN:575, 639-660, 672, 716-741."""
        y = dict(y or {})
        rew = y.get("reward", {}) or {}
        self.pairing_threshold_quantile = float(rew.get("pairing_threshold_quantile", self.pairing_threshold_quantile))
        self.mask_enable = bool(rew.get("mask_enable", self.mask_enable))
        self.mask_topk_start = int(rew.get("mask_topk_start", self.mask_topk_start))
        self.mask_topk_end = int(rew.get("mask_topk_end", self.mask_topk_end))
        self.mask_tau_q_start = float(rew.get("mask_tau_q_start", self.mask_tau_q_start))
        self.mask_tau_q_end = float(rew.get("mask_tau_q_end", self.mask_tau_q_end))
        self.mask_warmup_episodes = int(rew.get("mask_warmup_episodes", self.mask_warmup_episodes))
        self.mwm_allow_singles = bool(y.get("mwm_allow_singles", self.mwm_allow_singles))
        self.use_mwm_primary = bool(y.get("use_mwm_primary", self.use_mwm_primary))
        self.mwm_accept_quantile = float(y.get("mwm_accept_quantile", self.mwm_accept_quantile))
        self.min_pair_target = int(y.get("min_pair_target", self.min_pair_target))
        self.mwm_backoff_rounds = int(y.get("mwm_backoff_rounds", self.mwm_backoff_rounds))
        self.mwm_accept_q_step = float(y.get("mwm_accept_q_step", self.mwm_accept_q_step))
        self.qos_enable = bool(y.get("qos_enable", True))                 # TRAIN:672 (driver default True)
        self.qos_R_min_bpsHz = float(y.get("qos_R_min_bpsHz", 0.15))      # TRAIN:505-508 Config value
        self.score_w_delta_db = float(y.get("score_w_delta_db", 1.0))
        self.score_w_history = float(y.get("score_w_history", 0.3))
        self.pair_hist_decay = float(y.get("pair_hist_decay", 0.97))
        self.abs_gain_min_db = float(y.get("abs_gain_min_db", -math.inf))
        self.relax_q_step = float(y.get("relax_q_step", 0.02))
        self.relax_topk_step = int(y.get("relax_topk_step", 1))
        self.relax_tau_factor_per_round = float(y.get("relax_tau_factor_per_round", 0.95))
        if "max_backoff_rounds" in y and "mwm_backoff_rounds" not in y:   # TRAIN:731-732
            self.mwm_backoff_rounds = int(y["max_backoff_rounds"])
        self.freeze_group_in_episode = bool(y.get("freeze_group_in_episode", True))
        self.freeze_recalc_every = int(y.get("freeze_recalc_every", 0))
        self.freeze_unstick_prob = float(y.get("freeze_unstick_prob", 0.0))
        self.freeze_reward_drop_ratio = float(y.get("freeze_reward_drop_ratio", 0.05))
        return self

    def mask_schedule(self, i_episode: int) -> Tuple[float, int]:
        """(q_now, K_now), TRAIN:1323-1332."""
        prog = min(1.0, i_episode / max(1, self.mask_warmup_episodes))
        K = anneal_topk(i_episode, self.n_veh, self.mask_topk_start, self.mask_topk_end, self.mask_warmup_episodes)
        q = float(self.mask_tau_q_start + (self.mask_tau_q_end - self.mask_tau_q_start) * prog)
        return q, K

    def to_c(self, noise_power: float, P_max: float) -> N.RisVecNomaParams:
        if not self.use_mwm_primary:
            raise NotImplementedError("only the driver's default pairing path (use_mwm_primary) is built; "
                                      "the legacy mutual/greedy path (TRAIN:196-258) is out of scope")
        p = N.RisVecNomaParams()
        p.min_pair_target = max(1, int(self.min_pair_target))
        p.mwm_backoff_rounds = int(self.mwm_backoff_rounds)
        p.mwm_allow_singles = int(bool(self.mwm_allow_singles))
        p.qos_enable = int(bool(self.qos_enable))
        p.relax_topk_step = int(self.relax_topk_step)
        p.freeze_group_in_episode = int(bool(self.freeze_group_in_episode))
        p.freeze_recalc_every = int(self.freeze_recalc_every)
        p.mask_enable = int(bool(self.mask_enable))
        p.mwm_accept_quantile = float(self.mwm_accept_quantile)
        p.mwm_accept_q_step = float(self.mwm_accept_q_step)
        p.completion_min_quantile = float(self.completion_min_quantile)
        p.score_w_delta_db = float(self.score_w_delta_db)
        p.abs_gain_min_db = float(self.abs_gain_min_db)
        p.qos_soft_penalty = float(self.qos_soft_penalty_dbscore)
        p.qos_R_min = float(self.qos_R_min_bpsHz)
        p.noise_power = float(noise_power)
        p.P_max = float(P_max)
        p.relax_tau_factor = float(self.relax_tau_factor_per_round)
        p.tau_back_floor_db = float(self.tau_back_floor_db)
        p.freeze_reward_drop_ratio = float(self.freeze_reward_drop_ratio)
        p.freeze_unstick_prob = float(self.freeze_unstick_prob)
        p.score_w_history = float(self.score_w_history)
        p.pair_hist_decay = float(self.pair_hist_decay)
        return p


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class NomaGrouper:
    """Episode-scoped pairing state of TRAIN:1282-1300 for E envs + the three driver moments."""

    def __init__(self, env, config: Optional[NomaConfig] = None):
        self.env = env
        self.n_envs, self.n_veh = int(env.n_envs), int(env.n_veh)
        if not 1 <= self.n_veh <= N.NOMA_MAX_VEH:
            raise ValueError("NOMA grouping supports 1..%d vehicles, got %d" % (N.NOMA_MAX_VEH, self.n_veh))
        self.config = config if config is not None else NomaConfig(self.n_veh)
        self.device = env.device
        self._t = {}
        self._cstate: Optional[N.RisVecNomaState] = None
        self.i_episode = 0
        self._mask_fresh = False
        self._have_mask = False
        self._have_reward = False
        self._q_now: Optional[float] = None
        self._K_now: Optional[int] = None
        self._calls = 0

    # ------------------------------------------------------------------ device state
    def _ensure_device(self) -> None:
        if self._cstate is not None:
            return
        N.load()
        if self.device.type != "cuda" or not torch.cuda.is_available():
            raise RuntimeError("ris_vec_marl_amd needs a HIP device; there is no CPU fallback")
        E, V, dev = self.n_envs, self.n_veh, self.device
        z = lambda *shape, dt: torch.zeros(*shape, dtype=dt, device=dev)   # noqa: E731
        t = self._t
        t["hist"] = z(E, V, V, dt=torch.float32)
        t["streak"] = z(E, V, dt=torch.int32)
        t["partner"] = z(E, V, dt=torch.int32)
        t["n_groups"] = z(E, dt=torch.int32)
        t["last_global"] = z(E, dt=torch.float64)
        t["best_global"] = z(E, dt=torch.float64)
        t["flags"] = z(E, dt=torch.uint8)
        t["mask"] = z(E, V, V, dt=torch.uint8)
        t["tau"] = z(E, dt=torch.float64)
        t["pending"] = z(E, dt=torch.int32)
        t["info"] = z(E, 4, dt=torch.int32)
        s = N.RisVecNomaState()
        s.n_envs, s.n_veh, s.env_offset = E, V, int(getattr(self.env, "env_offset", 0))
        for k in ("hist", "streak", "partner", "n_groups", "last_global", "best_global", "flags", "mask", "tau",
                  "pending"):
            setattr(s, k, t[k].data_ptr())
        n_scratch = int(N.load().risvec_noma_scratch_bytes(E, V))   # V > 8: the envs group()'s first launch leaves to its second
        if n_scratch > 0:
            t["scratch"] = torch.zeros(n_scratch, dtype=torch.uint8, device=dev)
            s.scratch, s.scratch_bytes = t["scratch"].data_ptr(), n_scratch
        self._cstate = s

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def _params(self) -> N.RisVecNomaParams:
        return self.config.to_c(float(self.env.noise_power), float(self.env.P_max))

    # ------------------------------------------------------------------ driver moments
    def begin_episode(self, i_episode: int = 0) -> None:
        """TRAIN:1282-1300: clear the pair history, streaks, frozen groups, reward tracking and the
        cached mask / (q, K, tau)."""
        self._ensure_device()
        self.i_episode = int(i_episode)
        N.check(N.load().risvec_noma_begin_episode(C.byref(self._cstate), self._stream()))
        self._mask_fresh = self._have_mask = self._have_reward = False
        self._q_now = self._K_now = None

    def refresh_mask(self, gain: Optional[torch.Tensor] = None, gdb15: Optional[torch.Tensor] = None
                     ) -> Optional[torch.Tensor]:
        """TRAIN:1319-1343 on a channel-refresh step: tau = quantile q_now of the strong-vs-weak dB
        gaps, mask = (gap >= tau) + per-row Top-K, symmetrised; cached for the back-off.  Returns the
        [E,N,N] uint8 mask (the policy's action mask, TRAIN:1373-1378), or None when masking is off."""
        self._ensure_device()
        if not self.config.mask_enable:
            return None
        q_now, K_now = self.config.mask_schedule(self.i_episode)
        g = self._gain(gain)
        N.check(N.load().risvec_noma_mask(C.byref(self._cstate), _ptr(g), _ptr(self._f64(gdb15)), q_now, K_now,
                                          self._stream()))
        self._q_now, self._K_now = q_now, K_now
        self._mask_fresh = self._have_mask = True
        return self._t["mask"]

    def group(self, p_off01: Optional[torch.Tensor], i_step: int, gain: Optional[torch.Tensor] = None,
              prev_global: Optional[torch.Tensor] = None, gdb12: Optional[torch.Tensor] = None,
              gdb15: Optional[torch.Tensor] = None, u_unstick: Optional[torch.Tensor] = None,
              power_raw: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """TRAIN:1401-1562 for every env -> (partner [E,N] int32, n_groups [E] int32), the batched
        `noma_groups` `VecEnviron.step` takes (views of the grouper's state, valid until the next call).
        `p_off01` [E,N] is the offload power in [0,1] the policy chose (TRAIN:1391-1396); alternatively
        `power_raw` [E,N,2], the SAC power head itself in [-1,1], mapped in-kernel exactly as `marshal_actions`
        maps it (`risvec_noma_group_raw`).  `prev_global` defaults to the global reward the env's last
        `step` left in `metrics[:,0]` (none before the first step of the episode).  `gdb12` / `gdb15`
        inject a host's float64 dB gains (parity interface); `u_unstick` injects the TRAIN:1539 draw."""
        self._ensure_device()
        cfg, t = self.config, self._t
        g = self._gain(gain)
        if prev_global is None and self._have_reward:
            prev_global = self.env._t["metrics"]
        stride = 1
        if prev_global is not None:
            if prev_global.dim() == 2:
                stride = prev_global.stride(0)
            if prev_global.dtype != torch.float32 or prev_global.device != self.device:
                raise ValueError("prev_global must be a float32 tensor on %s" % self.device)
        if cfg.mask_enable and self._have_mask:                        # TRAIN:1486-1491
            K_back, tau_back = self._K_now, t["tau"]
        else:
            q_back = float(cfg.pairing_threshold_quantile)
            K_back = anneal_topk(self.i_episode, self.n_veh, cfg.mask_topk_start, cfg.mask_topk_end,
                                 cfg.mask_warmup_episodes)
            N.check(N.load().risvec_noma_mask(C.byref(self._cstate), _ptr(g), _ptr(self._f64(gdb15)), q_back, 0,
                                              self._stream()))
            tau_back = t["tau"]
        p01 = None
        fn = N.load().risvec_noma_group
        if power_raw is not None:
            if p_off01 is not None:
                raise ValueError("give p_off01 or power_raw, not both")
            p01 = power_raw.to(self.device, torch.float32).contiguous()
            if tuple(p01.shape) != (self.n_envs, self.n_veh, 2):
                raise ValueError("power_raw must have shape [n_envs, n_veh, 2]")
            fn = N.load().risvec_noma_group_raw
        elif p_off01 is not None:
            p01 = p_off01.to(self.device, torch.float32).contiguous()
            if tuple(p01.shape) != (self.n_envs, self.n_veh):
                raise ValueError("p_off01 must have shape [n_envs, n_veh]")
        uu = None if u_unstick is None else u_unstick.to(self.device, torch.float32).contiguous()
        self._calls += 1
        N.check(fn(
            C.byref(self._cstate), C.byref(self._params()), _ptr(g), _ptr(self._f64(gdb12)), _ptr(p01),
            1 if (cfg.mask_enable and self._mask_fresh) else 0, int(K_back), _ptr(tau_back),
            _ptr(prev_global), int(stride), int(i_step), _ptr(uu), int(getattr(self.env, "seed", 0)),
            self._calls, _ptr(t["info"]), self._stream()))
        self._mask_fresh = False
        self._have_reward = True        # the caller steps the env next; its metrics[:,0] feeds the next call
        return t["partner"], t["n_groups"]

    def bind_group(self, p_off01: Optional[torch.Tensor] = None, power_raw: Optional[torch.Tensor] = None):
        """`group()` with everything that does not change from step to step validated and
        marshalled once: returns `launch(i_step)`, one pre-built C-ABI call per step (the frozen
        steps of an episode are launch-bound, so host time matters).  Inputs are read in place:
        `p_off01` (updated by the caller between steps), the env's `gain` and the global reward
        its last `step` left in `metrics[:,0]`; `info` is not refreshed on this path.  Valid for the current episode's mask state and
        config; `begin_episode` / `refresh_mask` / config changes need a new binding only if
        `mask_enable` is off (tau is then recomputed per call by `group()`, not here)."""
        self._ensure_device()
        cfg, t = self.config, self._t
        if not cfg.mask_enable:
            raise ValueError("bind_group needs mask_enable (the cached tau / K of the last refresh_mask)")
        p01 = None
        lib, cs, prm = N.load(), C.byref(self._cstate), self._params()
        fn, check = lib.risvec_noma_group, N.check
        if power_raw is not None:                     # the SAC power head itself, mapped in-kernel (no marshalling launch)
            if p_off01 is not None:
                raise ValueError("give p_off01 or power_raw, not both")
            if (power_raw.dtype != torch.float32 or power_raw.device != self.device or not power_raw.is_contiguous()
                    or tuple(power_raw.shape) != (self.n_envs, self.n_veh, 2)):
                raise ValueError("power_raw must be a contiguous float32 [n_envs, n_veh, 2] tensor on %s" % self.device)
            p01, fn = power_raw, lib.risvec_noma_group_raw
        elif p_off01 is not None:
            if (p_off01.dtype != torch.float32 or p_off01.device != self.device or not p_off01.is_contiguous()
                    or tuple(p_off01.shape) != (self.n_envs, self.n_veh)):
                raise ValueError("p_off01 must be a contiguous float32 [n_envs, n_veh] tensor on %s" % self.device)
            p01 = p_off01
        g, metrics = self.env._t["gain"], self.env._t["metrics"]
        gp, pp, tp, mp, ip = g.data_ptr(), _ptr(p01), t["tau"].data_ptr(), metrics.data_ptr(), t["info"].data_ptr()
        stride, seed, stream = int(metrics.stride(0)), int(getattr(self.env, "seed", 0)), self._stream()
        pref = C.byref(prm)

        def launch(i_step: int) -> None:
            if not self._have_mask:
                raise RuntimeError("bind_group: call refresh_mask() first in each episode")
            self._calls += 1
            check(fn(cs, pref, gp, None, pp, 1 if self._mask_fresh else 0, self._K_now, tp,
                     mp if self._have_reward else None, stride, i_step, None, seed, self._calls, None, stream))
            self._mask_fresh = False
            self._have_reward = True

        launch.keepalive = (prm, p01, g, metrics)
        return launch

    def flush(self) -> None:
        """Apply the frozen steps still owed to `pair_affinity_hist` / `unpaired_streak` (the kernels
        defer that bookkeeping and replay it operation for operation when it is next needed)."""
        self._ensure_device()
        N.check(N.load().risvec_noma_flush(C.byref(self._cstate), C.byref(self._params()), self._stream()))

    # ------------------------------------------------------------------ views
    def _gain(self, gain):
        g = self.env._t["gain"] if gain is None else gain.to(self.device, torch.float32).contiguous()
        if tuple(g.shape) != (self.n_envs, self.n_veh):
            raise ValueError("gain must have shape [n_envs, n_veh]")
        return g

    def _f64(self, x):
        if x is None:
            return None
        x = x.to(self.device, torch.float64).contiguous()
        if tuple(x.shape) != (self.n_envs, self.n_veh):
            raise ValueError("dB gains must have shape [n_envs, n_veh]")
        return x

    @property
    def mask(self) -> torch.Tensor:
        return self._t["mask"]

    @property
    def tau(self) -> torch.Tensor:
        return self._t["tau"]

    @property
    def pair_affinity_hist(self) -> torch.Tensor:
        self.flush()
        return self._t["hist"]

    @property
    def unpaired_streak(self) -> torch.Tensor:
        self.flush()
        return self._t["streak"]

    @property
    def info(self) -> torch.Tensor:
        """[E,4] int32: recomputed?, back-off rounds, pairs, matchable users of the last matching."""
        return self._t["info"]

    @property
    def flags(self) -> torch.Tensor:
        return self._t["flags"]

    # ------------------------------------------------------------------ reference-shaped view, checkpoint
    def groups_list(self, e: int = 0):
        """`noma_groups` of env `e` as the reference builds it (TRAIN:1552-1553): pairs `[i, j]` (i < j)
        in increasing order, then the singles `[k]`.  Host copy; for the E=1 facade / debugging."""
        p = self._t["partner"][e].cpu().tolist()
        pairs = sorted([i, q] for i, q in enumerate(p) if 0 <= q < (1 << 16))
        return pairs + [[k] for k, q in enumerate(p) if q == -1]

    _STATE_KEYS = ("hist", "streak", "partner", "n_groups", "last_global", "best_global", "flags", "mask", "tau")

    def state_dict(self) -> dict:
        """Episode-scoped pairing state (deferred bookkeeping applied first), host tensors."""
        self.flush()
        sd = {k: self._t[k].detach().cpu().clone() for k in self._STATE_KEYS}
        sd["scalars"] = dict(i_episode=self.i_episode, mask_fresh=self._mask_fresh, have_mask=self._have_mask,
                             have_reward=self._have_reward, q_now=self._q_now, K_now=self._K_now, calls=self._calls)
        return sd

    def load_state_dict(self, sd: dict) -> None:
        self._ensure_device()
        for k in self._STATE_KEYS:
            self._t[k].copy_(sd[k].to(self.device))
        self._t["pending"].zero_()
        s = sd["scalars"]
        self.i_episode, self._mask_fresh, self._have_mask = int(s["i_episode"]), bool(s["mask_fresh"]), bool(s["have_mask"])
        self._have_reward, self._q_now, self._K_now, self._calls = bool(s["have_reward"]), s["q_now"], s["K_now"], int(s["calls"])
