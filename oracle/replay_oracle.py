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


class ReplayOracle:
    """BUF:3-37 with batched stores."""

    def __init__(self, max_size: int, input_shape: int, n_actions: int, n_agents: int):
        self.mem_size = int(max_size)                                                         # BUF:5
        self.mem_cntr = 0                                                                     # BUF:6
        self.state_memory = np.zeros((self.mem_size, input_shape * n_agents), np.float32)     # BUF:7
        self.action_memory = np.zeros((self.mem_size, n_actions * n_agents), np.float32)      # BUF:8
        self.reward_global_memory = np.zeros(self.mem_size, np.float32)                       # BUF:9
        self.reward_local_memory = np.zeros((self.mem_size, n_agents), np.float32)            # BUF:10
        self.new_state_memory = np.zeros((self.mem_size, input_shape * n_agents), np.float32)  # BUF:11
        self.terminal_memory = np.zeros(self.mem_size, dtype=bool)                            # BUF:12
        self.mask_memory = np.zeros((self.mem_size, n_agents * n_agents), np.float32)         # BUF:14

    def store_batch(self, state, action, reward_g, reward_l, state_, done, mask_flat) -> None:
        """E x BUF:16-25, env 0 first.  `done` scalar or [E]; mask_flat None = all ones (TRAIN:1786-1787)."""
        E = len(state)
        done = np.broadcast_to(np.asarray(done, dtype=bool), (E,))
        if mask_flat is None:
            mask_flat = np.ones((E, self.mask_memory.shape[1]), np.float32)
        for e in range(E):
            i = self.mem_cntr % self.mem_size
            self.state_memory[i] = state[e]
            self.action_memory[i] = action[e]
            self.reward_global_memory[i] = reward_g[e]
            self.reward_local_memory[i] = reward_l[e]
            self.new_state_memory[i] = state_[e]
            self.terminal_memory[i] = done[e]
            self.mask_memory[i] = mask_flat[e]
            self.mem_cntr += 1

    def sample(self, batch: np.ndarray) -> Tuple[np.ndarray, ...]:
        """BUF:27-37 with the drawn indices given (the reference draws np.random.choice(max_mem, n))."""
        return (self.state_memory[batch], self.action_memory[batch], self.reward_global_memory[batch],
                self.reward_local_memory[batch], self.new_state_memory[batch], self.terminal_memory[batch],
                self.mask_memory[batch])

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
