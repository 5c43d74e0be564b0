#!/usr/bin/env python3
"""Capture what the REFERENCE does in the two places where the E=1 facade (`ris_vec_marl_amd/compat.py`) has to
make a choice -- runs only in the build container, imports
/root/reference/Simulation-MARL-BCD/Environment.py read-only, writes data-only fixtures:

  tests/golden/facade_groups_8.npz   noma_groups lists the driver never builds but `step` accepts (Environment.py:339-369):
      a vehicle listed in several groups (the LAST group that lists it decides its rate; a partner that was paired
      with it earlier keeps the pair rate), pairs [u, u], groups of 3+ members and empty groups (ignored, but counted
      in G = len(noma_groups)).  Inputs + the reference's outputs for 256 random samples.
  tests/golden/facade_alias_8.npz    the returned arrays are live aliases of env state (Environment.py:731): what the
      tuple returned by step t reads AFTER step t+1 has run, next to what it read when it was returned.
"""
from __future__ import annotations

import os
import random
import sys

import numpy as np

REF_DIR = "/root/reference/Simulation-MARL-BCD"
OUT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
if not os.path.isfile(os.path.join(REF_DIR, "Environment.py")):
    sys.exit("reference not present at %s (this tool only runs in the build container)" % REF_DIR)
sys.dont_write_bytecode = True
sys.path.insert(0, REF_DIR)
import Environment as REF  # noqa: E402  (the reference itself)

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import risvec_oracle as orc  # noqa: E402


def make_env(V, M=16):
    L = orc.default_lanes()
    env = REF.Environ(L["down"], L["up"], L["left"], L["right"], 400, 400, V, M, 3)
    p = orc.OracleParams.yaml_effective()
    env.bandwidth = p.bandwidth; env.noise_power = p.noise_power; env.P_max = p.P_max
    env.f_local_max = p.f_local_max; env.cycles_per_bit = p.cycles_per_bit; env.rate = p.rate
    env.w_d, env.w_e = p.w_d, p.w_e
    env.R_min_bpsHz, env.D_max_s, env.qos_penalty = p.R_min_bpsHz, p.D_max_s, p.qos_penalty
    return env


def quirky_groups(V, rng):
    """A random noma_groups list with the shapes the driver never produces."""
    groups = []
    for _ in range(int(rng.integers(1, V + 3))):
        kind = rng.choice(["single", "pair", "same", "triple", "empty", "quad"], p=[0.3, 0.35, 0.08, 0.12, 0.1, 0.05])
        if kind == "single":
            groups.append([int(rng.integers(0, V))])
        elif kind == "pair":
            a, b = rng.choice(V, 2, replace=False)
            groups.append([int(a), int(b)])
        elif kind == "same":
            u = int(rng.integers(0, V))
            groups.append([u, u])
        elif kind == "triple":
            groups.append([int(x) for x in rng.choice(V, 3, replace=False)])
        elif kind == "quad":
            groups.append([int(x) for x in rng.choice(V, 4, replace=False)])
        else:
            groups.append([])
    return groups


def capture_groups(V=8, n=256, seed=123):
    rng = np.random.default_rng(seed)
    np.random.seed(seed); random.seed(seed)
    GMAX, LMAX = V + 3, 4
    out = dict(data_buf0=np.zeros((n, V)), mec_q0=np.zeros(n), gain=np.zeros((n, V)), action=np.zeros((n, 2, V)),
               groups=np.full((n, GMAX, LMAX), -1, dtype=np.int64), group_len=np.full((n, GMAX), -1, dtype=np.int64),
               arrivals=np.zeros((n, V), dtype=np.int64), reward=np.zeros((n, V)), global_reward=np.zeros(n),
               data_buf=np.zeros((n, V)), data_t=np.zeros((n, V)), data_p=np.zeros((n, V)), vehicle_rate=np.zeros((n, V)),
               mec_q=np.zeros(n))
    env = make_env(V)
    env.make_new_game()
    n_dup = 0
    for i in range(n):
        B0 = rng.uniform(0, 10, V); q0 = float(rng.uniform(0, 4e6)) if rng.random() < 0.5 else 0.0
        gain = 10 ** rng.uniform(-13, -10, V)
        if rng.random() < 0.15:
            gain[rng.integers(0, V)] = gain[rng.integers(0, V)]          # exact gain ties
        act = rng.uniform(-0.1, 1.2, (2, V))
        groups = quirky_groups(V, rng)
        flat = [u for g in groups if len(g) in (1, 2) for u in g]
        n_dup += int(len(flat) != len(set(flat)))
        env.DataBuf = B0.copy(); env.mec_queue_cycles = q0; env.channel_gains = gain.copy()
        r = env.step(act.copy(), [list(g) for g in groups])
        out["data_buf0"][i], out["mec_q0"][i], out["gain"][i], out["action"][i] = B0, q0, gain, act
        for k, g in enumerate(groups):
            out["group_len"][i, k] = len(g)
            out["groups"][i, k, :len(g)] = g
        out["arrivals"][i] = np.asarray(env.data_r, dtype=np.int64)
        out["reward"][i], out["global_reward"][i] = r[0], r[1]
        out["data_buf"][i], out["data_t"][i], out["data_p"][i] = r[2], r[3], r[4]
        out["vehicle_rate"][i] = env.vehicle_rate
        out["mec_q"][i] = env.mec_queue_cycles
    assert n_dup > n // 4, n_dup
    np.savez_compressed(os.path.join(OUT_DIR, "facade_groups_%d.npz" % V), **out)
    print("facade_groups_%d.npz: %d samples, %d with a vehicle in more than one scheduled group" % (V, n, n_dup))


def capture_alias(V=8, T=4, seed=7):
    rng = np.random.default_rng(seed)
    np.random.seed(seed); random.seed(seed)
    env = make_env(V)
    env.make_new_game()
    env.channel_gains = 10 ** rng.uniform(-12, -10, V)
    groups = [[0, 1], [2], [3], [4, 5], [6], [7]]
    names = ("data_buf", "data_t", "data_p")
    at_return = {k: np.zeros((T, V)) for k in names}
    after_next = {k: np.zeros((T - 1, V)) for k in names}
    same_object = np.zeros((T - 1, 3), dtype=np.int64)
    actions = rng.uniform(0, 1, (T, 2, V))
    arrivals = np.zeros((T, V), dtype=np.int64)
    buf0 = env.DataBuf.copy()
    prev = None
    for t in range(T):
        r = env.step(actions[t].copy(), groups)
        arrivals[t] = np.asarray(env.data_r, dtype=np.int64)
        if prev is not None:
            for j, k in enumerate(names):
                after_next[k][t - 1] = prev[j]                      # the OLD tuple, read now
                same_object[t - 1, j] = int(prev[j] is (r[2], r[3], r[4])[j])
        for j, k in enumerate(names):
            at_return[k][t] = (r[2], r[3], r[4])[j]
        prev = (r[2], r[3], r[4])
    np.savez_compressed(os.path.join(OUT_DIR, "facade_alias_%d.npz" % V), gain=env.channel_gains, data_buf0=buf0,
                        actions=actions, arrivals=arrivals, same_object=same_object,
                        **{"ret_" + k: v for k, v in at_return.items()}, **{"later_" + k: v for k, v in after_next.items()})
    print("facade_alias_%d.npz: returned data_t/data_p are the same objects across steps: %s; DataBuf: %s"
          % (V, bool(same_object[:, 1:].all()), bool(same_object[:, 0].all())))


if __name__ == "__main__":
    capture_groups()
    capture_alias()
