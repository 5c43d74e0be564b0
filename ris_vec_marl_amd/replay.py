"""f3: the reference's replay buffer (`Simulation-MARL-BCD/buffer.py`, BUF below) kept in
HBM, and the marshalling its driver does around `env.step()` (`marl_train_bcd.py`, TRAIN).

`VecReplayBuffer` has buffer.py's constructor, attribute names and two methods
(`store_transition`, `sample_buffer`, BUF:16-37); `store_batch` appends the E transitions of one
vectorised step in a single launch, reading the step kernel's outputs in place (obs, reward,
metrics[:,0], the NOMA mask).  `marshal_actions` is TRAIN:1386-1396, 1601-1608, 1776-1784 for all
envs.  No CPU path: every method launches HIP kernels through the C ABI (`risvec_replay_*`,
`risvec_marshal_actions`).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np
import torch

from . import _native as N


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class VecReplayBuffer:
    """ReplayBuffer(max_size, input_shape, n_actions, n_agents) of BUF:3-14, arrays on `device`."""

    def __init__(self, max_size: int, input_shape: int, n_actions: int, n_agents: int, device="cuda", seed: int = 0):
        N.load()
        self.device = N.resolve_device(device)
        if self.device.type != "cuda" or not torch.cuda.is_available():
            raise RuntimeError("ris_vec_marl_amd needs a HIP device; there is no CPU fallback")
        self.mem_size = int(max_size)
        self.mem_cntr = 0
        self.input_shape, self.n_actions, self.n_agents = int(input_shape), int(n_actions), int(n_agents)
        self.seed = int(seed)
        self._samples = 0
        z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=self.device)   # noqa: E731
        S, A, L = self.input_shape * self.n_agents, self.n_actions * self.n_agents, self.n_agents
        self.state_memory = z(self.mem_size, S)
        self.action_memory = z(self.mem_size, A)
        self.reward_global_memory = z(self.mem_size)
        self.reward_local_memory = z(self.mem_size, L)
        self.new_state_memory = z(self.mem_size, S)
        self.terminal_memory = z(self.mem_size, dt=torch.bool)
        self.mask_memory = z(self.mem_size, L * L)
        rb = N.RisVecReplay()
        rb.n_agents, rb.input_shape, rb.n_actions, rb.mem_size = L, self.input_shape, self.n_actions, self.mem_size
        for k in ("state_memory", "action_memory", "reward_global_memory", "reward_local_memory", "new_state_memory",
                  "terminal_memory", "mask_memory"):
            setattr(rb, k, getattr(self, k).data_ptr())
        self._c = rb

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def _f32(self, x, shape, name):
        t = torch.as_tensor(x) if not isinstance(x, torch.Tensor) else x
        t = t.to(self.device, torch.float32).reshape(shape).contiguous()
        return t

    # ------------------------------------------------------------------ stores
    def store_batch(self, state: torch.Tensor, action: Optional[torch.Tensor], reward_g: torch.Tensor, reward_l: torch.Tensor,
                    state_: torch.Tensor, done=False, mask: Optional[torch.Tensor] = None,
                    policy_out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> None:
        """n consecutive store_transition calls (BUF:16-25), env 0 first.  state / state_ [n, ...]
        (any trailing shape with input_shape*n_agents elements, e.g. the env's obs [E,V,5]); action
        [n, n_actions*n_agents]; reward_g [n] or a strided [n, k] tensor whose column 0 is used
        (the env's metrics); reward_l [n, n_agents]; done: bool or [n] bool/uint8; mask [n, A, A]
        uint8/bool (the NOMA mask) or None = all ones (TRAIN:1786-1787).
        `action=None, policy_out=(power_raw [n,A,2], probs [n,A,A])`: the action row is built in the store kernel
        from the policy outputs (what `marshal_actions` would have written, TRAIN:1386-1390, 1776-1784)."""
        n = int(state.shape[0])
        S, A, L = self.input_shape * self.n_agents, self.n_actions * self.n_agents, self.n_agents
        st, st2 = self._f32(state, (n, S), "state"), self._f32(state_, (n, S), "state_")
        rl = self._f32(reward_l, (n, L), "reward_l")
        ac = pw = pr = None
        if action is not None:
            ac = self._f32(action, (n, A), "action")
        elif policy_out is None:
            raise ValueError("store_batch: give action or policy_out=(power_raw, probs)")
        else:
            pw, pr = self._f32(policy_out[0], (n, L, 2), "power_raw"), self._f32(policy_out[1], (n, L, L), "probs")
        rg = reward_g if isinstance(reward_g, torch.Tensor) else torch.as_tensor(reward_g)
        if rg.dtype != torch.float32 or rg.device != self.device:
            rg = rg.to(self.device, torch.float32)
        if rg.dim() == 2:
            stride = rg.stride(0)
        else:
            rg = rg.reshape(n).contiguous()
            stride = 1
        if rg.shape[0] != n:
            raise ValueError("reward_g must have n rows")
        dn, done_all = None, 0
        if isinstance(done, (bool, np.bool_, int)):
            done_all = int(bool(done))
        else:
            dn = torch.as_tensor(done).to(self.device).to(torch.uint8).reshape(n).contiguous()
        mk = None
        if mask is not None:
            mk = mask.to(self.device)
            mk = (mk != 0).to(torch.uint8).reshape(n, L * L).contiguous() if mk.dtype != torch.uint8 \
                else mk.reshape(n, L * L).contiguous()
        if ac is not None:
            N.check(N.load().risvec_replay_store(C.byref(self._c), self.mem_cntr, n, _ptr(st), _ptr(ac), _ptr(rg), int(stride),
                                                 _ptr(rl), _ptr(st2), _ptr(dn), done_all, _ptr(mk), None, self._stream()))
        else:
            N.check(N.load().risvec_replay_store_policy(C.byref(self._c), self.mem_cntr, n, _ptr(st), _ptr(pw), _ptr(pr),
                                                        _ptr(rg), int(stride), _ptr(rl), _ptr(st2), _ptr(dn), done_all,
                                                        _ptr(mk), None, self._stream()))
        self.mem_cntr += n

    def bind_store(self, state: Optional[torch.Tensor], action: Optional[torch.Tensor], reward_g: torch.Tensor,
                   reward_l: torch.Tensor, state_: torch.Tensor, mask: Optional[torch.Tensor] = None,
                   policy_out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
        """`store_batch` with the arguments validated and marshalled once: returns `launch(done=False,
        use_mask=True)`, one pre-built C-ABI call that appends the CURRENT contents of the given
        tensors (they are read in place every call: the env's obs / reward / metrics, the marshalled
        action rows, the NOMA mask).  All tensors must be contiguous float32 (mask uint8) on the device;
        reward_g may be the env's [E, k] metrics tensor (column 0 is read).
        `state=None`: the buffer carries the previous step's `state_` forward itself (the store kernel
        drops a copy of `state_` into a ping-pong buffer while it has it in registers), i.e. the
        driver's `marl_state_old_all = marl_state_new_all` (TRAIN:1277, 1774) without a copy kernel;
        the first call then stores the CURRENT `state_` as `state`.
        `action=None, policy_out=(power_raw, probs)`: the action row is built in the store kernel from the policy
        outputs, read in place (no marshalling launch; see `store_batch`)."""
        n = int(state_.shape[0])
        carry = None
        if state is None:
            carry = [state_.detach().clone().reshape(n, -1), torch.empty_like(state_).reshape(n, -1)]
            state = carry[0]
        S, A, L = self.input_shape * self.n_agents, self.n_actions * self.n_agents, self.n_agents

        def ok(t, numel, dt=torch.float32):
            return t.dtype == dt and t.device == self.device and t.is_contiguous() and t.numel() == numel
        if action is None:
            if policy_out is None or not (ok(policy_out[0], n * L * 2) and ok(policy_out[1], n * L * L)):
                raise ValueError("bind_store: without action, policy_out = (power_raw [n,A,2], probs [n,A,A]) contiguous "
                                 "float32 on %s is needed" % self.device)
            if self.n_actions != L + 2:
                raise ValueError("bind_store: the policy-output form needs n_actions = n_agents + 2")
        elif not ok(action, n * A):
            raise ValueError("bind_store: action must be a contiguous float32 [n, %d] tensor on %s" % (A, self.device))
        if not (ok(state, n * S) and ok(state_, n * S) and ok(reward_l, n * L)):
            raise ValueError("bind_store: state/reward_l/state_ must be contiguous float32 tensors of n rows on %s"
                             % self.device)
        if reward_g.dtype != torch.float32 or reward_g.device != self.device or reward_g.shape[0] != n:
            raise ValueError("bind_store: reward_g must be a float32 tensor with n rows on %s" % self.device)
        stride = int(reward_g.stride(0)) if reward_g.dim() == 2 else 1
        if mask is not None and not ok(mask, n * L * L, torch.uint8):
            raise ValueError("bind_store: mask must be a contiguous uint8 [n, A, A] tensor")
        lib, check, rb = N.load(), N.check, C.byref(self._c)
        if action is not None:
            fn_a, a_args = lib.risvec_replay_store, (action.data_ptr(),)
        else:
            fn_a, a_args = lib.risvec_replay_store_policy, (policy_out[0].data_ptr(), policy_out[1].data_ptr())
        ptrs = (state.data_ptr(), None, reward_g.data_ptr(), stride, reward_l.data_ptr(), state_.data_ptr())
        mp, stream = _ptr(mask), self._stream()

        flip = [0]

        def launch(done: bool = False, use_mask: bool = True) -> None:
            if carry is None:
                src, dst = ptrs[0], None
            else:
                src, dst = carry[flip[0]].data_ptr(), carry[flip[0] ^ 1].data_ptr()
                flip[0] ^= 1
            check(fn_a(rb, self.mem_cntr, n, src, *a_args, ptrs[2], ptrs[3], ptrs[4], ptrs[5], None,
                       1 if done else 0, mp if use_mask else None, dst, stream))
            self.mem_cntr += n

        launch.keepalive = (state, action, policy_out, reward_g, reward_l, state_, mask, carry)
        return launch

    def store_transition(self, state, action, reward_g, reward_l, state_, done, mask_flat) -> None:
        """BUF:16-25 with the reference's signature (one transition; NumPy arrays or tensors)."""
        def row(x):
            t = x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x, dtype=np.float32))
            return t.reshape(1, -1)
        L = self.n_agents
        mask = mask_flat if isinstance(mask_flat, torch.Tensor) else torch.as_tensor(np.asarray(mask_flat))
        self.store_batch(row(state), row(action), torch.tensor([float(reward_g)], dtype=torch.float32), row(reward_l),
                         row(state_), bool(done), (mask != 0).reshape(1, L, L))

    # ------------------------------------------------------------------ checkpoint
    _ARRAYS = ("state_memory", "action_memory", "reward_global_memory", "reward_local_memory", "new_state_memory",
               "terminal_memory", "mask_memory")

    def state_dict(self) -> dict:
        """The filled part of the ring + its counters (host tensors)."""
        n = min(self.mem_cntr, self.mem_size)
        sd = {k: getattr(self, k)[:n].detach().cpu().clone() for k in self._ARRAYS}
        sd["scalars"] = dict(mem_cntr=self.mem_cntr, mem_size=self.mem_size, samples=self._samples, seed=self.seed)
        return sd

    def load_state_dict(self, sd: dict) -> None:
        s = sd["scalars"]
        if int(s["mem_size"]) != self.mem_size:
            raise ValueError("replay checkpoint has mem_size %d, this buffer %d" % (s["mem_size"], self.mem_size))
        n = min(int(s["mem_cntr"]), self.mem_size)
        for k in self._ARRAYS:
            getattr(self, k)[:n].copy_(sd[k].to(self.device))
        self.mem_cntr, self._samples, self.seed = int(s["mem_cntr"]), int(s["samples"]), int(s["seed"])

    # ------------------------------------------------------------------ sampling
    def sample_buffer(self, batch_size: int, idx: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, ...]:
        """BUF:27-37 -> (states, actions, rewards_g, rewards_l, states_, dones, masks), device tensors.
        `idx` [batch] int64 injects the rows (parity with the reference's np.random.choice draw);
        otherwise rows are drawn by Philox on the device (`last_batch` holds them afterwards)."""
        B = int(batch_size)
        max_mem = min(self.mem_cntr, self.mem_size)
        if max_mem < 1:
            raise ValueError("sample_buffer on an empty buffer")
        S, A, L = self.input_shape * self.n_agents, self.n_actions * self.n_agents, self.n_agents
        dev = self.device
        out = (torch.empty(B, S, device=dev), torch.empty(B, A, device=dev), torch.empty(B, device=dev),
               torch.empty(B, L, device=dev), torch.empty(B, S, device=dev),
               torch.empty(B, dtype=torch.bool, device=dev), torch.empty(B, L * L, device=dev))
        ix = None
        if idx is not None:
            ix = idx.to(dev, torch.int64).contiguous()
            if ix.numel() != B:
                raise ValueError("idx must hold batch_size rows")
            if int(ix.min()) < 0 or int(ix.max()) >= max_mem:
                raise ValueError("idx outside [0, %d)" % max_mem)
        self.last_batch = torch.empty(B, dtype=torch.int64, device=dev)
        self._samples += 1
        N.check(N.load().risvec_replay_sample(C.byref(self._c), max_mem, B, _ptr(ix), self.seed, self._samples,
                                              *(t.data_ptr() for t in out), self.last_batch.data_ptr(), self._stream()))
        return out


def marshal_actions(power_raw: torch.Tensor, probs: Optional[torch.Tensor], cpu_share_floor: float = 0.10,
                    want_store: bool = True, out: Optional[tuple] = None
                    ) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
    """Policy outputs of all agents of all envs -> (action_env [E,2,V], p_off01 [E,V], action_store
    [E, V*(V+2)] or None): TRAIN:1601-1608 (env action), TRAIN:1391-1396 (pairing power) and
    TRAIN:1386-1390 + 1776-1784 (replay action row = per agent [probs_i with zero diagonal, raw
    power_i]).  power_raw [E,V,2] float32 in [-1,1]; probs [E,V,V] float32.  `out` = preallocated
    (action_env, p_off01, action_store) to write into."""
    N.load()
    if power_raw.device.type != "cuda":
        raise RuntimeError("ris_vec_marl_amd needs a HIP device; there is no CPU fallback")
    E, V = int(power_raw.shape[0]), int(power_raw.shape[1])
    pr = power_raw.to(torch.float32).contiguous()
    if tuple(pr.shape) != (E, V, 2):
        raise ValueError("power_raw must have shape [E, V, 2]")
    pb = None
    if want_store:
        if probs is None or tuple(probs.shape) != (E, V, V):
            raise ValueError("probs must have shape [E, V, V]")
        pb = probs.to(pr.device, torch.float32).contiguous()
    dev = pr.device
    if out is not None:             # preallocated (action_env, p_off01, action_store): stable pointers for bound launches
        action_env, p01, store = out
        for t, shape in ((action_env, (E, 2, V)), (p01, (E, V))) + (((store, (E, V * (V + 2))),) if want_store else ()):
            if t.dtype != torch.float32 or t.device != dev or not t.is_contiguous() or tuple(t.shape) != shape:
                raise ValueError("marshal_actions: out tensors must be contiguous float32 of shapes "
                                 "[E,2,V], [E,V], [E,V*(V+2)] on %s" % dev)
    else:
        action_env = torch.empty(E, 2, V, device=dev)
        p01 = torch.empty(E, V, device=dev)
        store = torch.empty(E, V * (V + 2), device=dev) if want_store else None
    N.check(N.load().risvec_marshal_actions(E, V, pr.data_ptr(), _ptr(pb), float(cpu_share_floor), action_env.data_ptr(),
                                            p01.data_ptr(), _ptr(store), torch.cuda.current_stream(dev).cuda_stream))
    return action_env, p01, store
