"""CPU ORACLE for the replay ring buffer + action marshalling -- TEST
INFRASTRUCTURE ONLY.

NumPy restatement of `Simulation-MARL-BCD/buffer.py` (BUF below) and of the marshalling the
driver `marl_train_bcd.py` (TRAIN) performs around `env.step()`: the policy-output -> env-action
map (TRAIN:1391-1396, 1601-1608) and the `[probs_i, power_i]` concatenation written to the replay
(TRAIN:1386-1390, 1776-1784), batched over E envs.  Storing the E transitions of one vectorised
step is defined as E consecutive `store_transition` calls in env order.

    Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may
    import it.  The product package must not (and does not).

Parity status: PINNED by `tests/golden/replay_*.npz` (tools/capture_golden_replay.py drives the
reference's own ReplayBuffer, recording the indices `np.random.choice` returned).
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def ring_layout(input_shape: int, n_actions: int, n_agents: int):
    """(array name, row shape, dtype) of the seven ring arrays, in the order `sample_buffer` returns
    them (BUF:7-14, 29-37)."""
    S, A, L = input_shape * n_agents, n_actions * n_agents, n_agents
    return (("state_memory", (S,), np.float32), ("action_memory", (A,), np.float32),
            ("reward_global_memory", (), np.float32), ("reward_local_memory", (L,), np.float32),
            ("new_state_memory", (S,), np.float32), ("terminal_memory", (), bool),
            ("mask_memory", (L * L,), np.float32))


class ReplayOracle:
    """BUF:3-37 with batched stores: row `mem_cntr % mem_size` of every array takes the next
    transition (BUF:17-25); sampling indexes all arrays with the same drawn rows (BUF:27-37)."""

    def __init__(self, max_size: int, input_shape: int, n_actions: int, n_agents: int):
        self.mem_size, self.mem_cntr = int(max_size), 0
        self.layout = ring_layout(input_shape, n_actions, n_agents)
        for name, tail, dtype in self.layout:
            setattr(self, name, np.zeros((self.mem_size,) + tail, dtype=dtype))

    def store_batch(self, state, action, reward_g, reward_l, state_, done, mask_flat) -> None:
        """E consecutive stores, env 0 first.  `done` scalar or [E]; mask_flat None = all ones
        (TRAIN:1786-1787)."""
        E = len(state)
        if mask_flat is None:
            mask_flat = np.ones((E, self.mask_memory.shape[1]), np.float32)
        columns = (state, action, reward_g, reward_l, state_, np.broadcast_to(np.asarray(done, dtype=bool), (E,)),
                   mask_flat)
        for e in range(E):
            row = self.mem_cntr % self.mem_size
            for (name, _, _), values in zip(self.layout, columns):
                getattr(self, name)[row] = values[e]
            self.mem_cntr += 1

    def sample(self, batch: np.ndarray) -> Tuple[np.ndarray, ...]:
        """The rows `batch` of every array (the reference draws them with np.random.choice(max_mem, n))."""
        return tuple(getattr(self, name)[batch] for name, _, _ in self.layout)

    def max_mem(self) -> int:
        return min(self.mem_cntr, self.mem_size)                                              # BUF:28


def marshal_actions(power_raw: np.ndarray, probs: np.ndarray, floor: float
                    ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """power_raw [E,V,2] float32 policy outputs, probs [E,V,V] float32 intent probabilities ->
      action_env [E,2,V] float64   TRAIN:1601-1608: clip to +-0.999 (float32), (x+1)/2 (float32 result stored
                                   into a float64 array), CPU share floored at clamp(floor, 0, 0.95)
      p_off01    [E,V]   float64   TRAIN:1391-1396 (same map, row 0)
      store      [E,V*(V+2)] f32   TRAIN:1386-1390, 1776-1784: per agent [probs_i with a zero diagonal, raw power_i]."""
    power_raw = np.asarray(power_raw, np.float32)
    E, V, _ = power_raw.shape
    clipped = np.clip(power_raw, np.float32(-0.999), np.float32(0.999))
    mapped = ((clipped + np.float32(1)) / np.float32(2)).astype(np.float64)       # float32 arithmetic, widened on store
    action_env = np.zeros((E, 2, V), np.float64)
    action_env[:, 0, :] = mapped[:, :, 0]
    fl = max(0.0, min(float(floor), 0.95))
    action_env[:, 1, :] = np.maximum(mapped[:, :, 1], fl)
    p = np.array(probs, np.float32, copy=True)
    p[:, np.arange(V), np.arange(V)] = 0.0
    store = np.concatenate([p, power_raw], axis=2).reshape(E, V * (V + 2)).astype(np.float32)
    return action_env, action_env[:, 0, :].copy(), store


def philox_sample_indices(n: int, max_mem: int, counter: int, seed: int) -> np.ndarray:
    """Production index draw of the build: Philox(seed; b, 0, counter, site 8) -> floor(x * max_mem / 2^32)."""
    from .risvec_oracle import philox4x32, randint_from_u32
    x = philox4x32(np.arange(n, dtype=np.uint64), np.zeros(n, np.uint64), np.full(n, counter, np.uint64),
                   np.full(n, 8, np.uint64), seed)[0]
    return randint_from_u32(x, 0, max_mem)
