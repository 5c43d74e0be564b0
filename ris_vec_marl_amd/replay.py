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
        self.device = torch.device(device)
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
    def store_batch(self, state: torch.Tensor, action: torch.Tensor, reward_g: torch.Tensor, reward_l: torch.Tensor,
                    state_: torch.Tensor, done=False, mask: Optional[torch.Tensor] = None) -> None:
        """n consecutive store_transition calls (BUF:16-25), env 0 first.  state / state_ [n, ...]
        (any trailing shape with input_shape*n_agents elements, e.g. the env's obs [E,V,5]); action
        [n, n_actions*n_agents]; reward_g [n] or a strided [n, k] tensor whose column 0 is used
        (the env's metrics); reward_l [n, n_agents]; done: bool or [n] bool/uint8; mask [n, A, A]
        uint8/bool (the NOMA mask) or None = all ones (TRAIN:1786-1787)."""
        n = int(state.shape[0])
        S, A, L = self.input_shape * self.n_agents, self.n_actions * self.n_agents, self.n_agents
        st, st2 = self._f32(state, (n, S), "state"), self._f32(state_, (n, S), "state_")
        ac, rl = self._f32(action, (n, A), "action"), self._f32(reward_l, (n, L), "reward_l")
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
        N.check(N.load().risvec_replay_store(C.byref(self._c), self.mem_cntr, n, _ptr(st), _ptr(ac), _ptr(rg), int(stride),
                                             _ptr(rl), _ptr(st2), _ptr(dn), done_all, _ptr(mk), self._stream()))
        self.mem_cntr += n

    def store_transition(self, state, action, reward_g, reward_l, state_, done, mask_flat) -> None:
        """BUF:16-25 with the reference's signature (one transition; NumPy arrays or tensors)."""
        L = self.n_agents
        mk = torch.as_tensor(np.asarray(mask_flat) != 0 if not isinstance(mask_flat, torch.Tensor) else mask_flat != 0)
        self.store_batch(torch.as_tensor(np.asarray(state, dtype=np.float32))[None] if not isinstance(state, torch.Tensor) else state[None],
                         torch.as_tensor(np.asarray(action, dtype=np.float32))[None] if not isinstance(action, torch.Tensor) else action[None],
                         torch.tensor([float(reward_g)], dtype=torch.float32),
                         torch.as_tensor(np.asarray(reward_l, dtype=np.float32))[None] if not isinstance(reward_l, torch.Tensor) else reward_l[None],
                         torch.as_tensor(np.asarray(state_, dtype=np.float32))[None] if not isinstance(state_, torch.Tensor) else state_[None],
                         bool(done), mk.reshape(1, L, L))

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
                    want_store: bool = True) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
    """Policy outputs of all agents of all envs -> (action_env [E,2,V], p_off01 [E,V], action_store
    [E, V*(V+2)] or None): TRAIN:1601-1608 (env action), TRAIN:1391-1396 (pairing power) and
    TRAIN:1386-1390 + 1776-1784 (replay action row = per agent [probs_i with zero diagonal, raw
    power_i]).  power_raw [E,V,2] float32 in [-1,1]; probs [E,V,V] float32."""
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
    action_env = torch.empty(E, 2, V, device=dev)
    p01 = torch.empty(E, V, device=dev)
    store = torch.empty(E, V * (V + 2), device=dev) if want_store else None
    N.check(N.load().risvec_marshal_actions(E, V, pr.data_ptr(), _ptr(pb), float(cpu_share_floor), action_env.data_ptr(),
                                            p01.data_ptr(), _ptr(store), torch.cuda.current_stream(dev).cuda_stream))
    return action_env, p01, store
