#!/usr/bin/env python3
"""Capture golden vectors for the per-episode metrics sink from the REFERENCE.
Runs ONLY in the CPU build container.

Steps the reference's own `Environ` (imported by tools/capture_golden.py) through episodes of random
actions / NOMA groups and keeps the episode sums with the driver's own statements
(marl_train_bcd.py:1611-1662, 1714-1753, 1769, 1824, 1838-1865, 1939-1941 -- `getattr(env,
"last_*")` added to Python floats step by step).  That script cannot be imported (it trains at import
and needs tensorboard), so its `_jain_index` helper is compiled from the script's syntax tree -- the
one function, nothing else of the file runs -- and called on the per-user means.  The fixture holds
data only: per-step metrics / rewards / powers as the reference env produced them and the episode
scalars.
"""
from __future__ import annotations

import ast
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import capture_golden as CG  # noqa: E402  (imports the reference Environment)

TRAIN = os.path.join(CG.REF_DIR, "marl_train_bcd.py")


def reference_function(name):
    tree = ast.parse(open(TRAIN, encoding="utf-8").read())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name)
    ns = {"np": np}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), TRAIN, "exec"), ns)
    return ns[name]


def capture(V, n_env, n_ep, n_step, seed):
    jain = reference_function("_jain_index")
    rec = dict(metrics=[], reward=[], power_w=[], episode=[])
    for e in range(n_env):
        np.random.seed(seed + e); random.seed(seed + e)
        rng = np.random.default_rng(seed + e)
        env = CG.make_env(V, 16)
        env.make_new_game()
        CG.apply_params(env, "yaml" if e % 2 == 0 else "default")
        m_env, r_env, p_env, ep_env = [], [], [], []
        for ep in range(n_ep):
            env.channel_gains = 10 ** rng.uniform(-13.0, -9.5, V)
            if e == 1 and ep == 0:
                env.DataBuf = rng.uniform(50.0, 300.0, V)            # heavy backlog: rewards below -5 get clipped
            # --- the driver's episode-start zeroing (TRAIN:1278-1300)
            ep_sum = {n: 0.0 for n in CG.METRICS}
            ep_steps = 0
            record_reward = np.zeros(V)
            Power, Power_local, Power_offload, raw_global = [], [], [], []
            last_env_global, ep_env_best = None, None
            ms, rs, ps = [], [], []
            for st in range(n_step):
                action = rng.uniform(0.0, 1.0, (2, V))
                action[1] = np.maximum(action[1], 0.1)
                groups = CG.random_groups(V, rng)
                per_user_reward, global_reward, _, _, _, _, _ = env.step(action, groups)
                _raw_global = float(global_reward)
                if last_env_global is None:                           # TRAIN:1613-1622
                    ep_env_best = _raw_global
                elif _raw_global > ep_env_best:
                    ep_env_best = _raw_global
                last_env_global = _raw_global
                raw_global.append(_raw_global)
                for n in CG.METRICS:                                  # TRAIN:1626-1662
                    v = getattr(env, n, None)
                    if v is not None:
                        ep_sum[n] += float(v)
                ep_steps += 1
                per_user_reward_norm = np.clip(np.array(per_user_reward).copy(), -5, 5)     # TRAIN:1711-1714
                _pw = getattr(env, "last_power_W", None)              # TRAIN:1717-1753
                Power.append(float(np.array(_pw).sum()))
                _pw_arr = np.array(_pw)
                Power_offload.append(float(_pw_arr[0, :].sum()))
                Power_local.append(float(_pw_arr[1, :].sum()))
                for i in range(V):                                    # TRAIN:1768-1769
                    record_reward[i] += per_user_reward_norm[i]
                ms.append([_raw_global] + [float(getattr(env, n)) for n in CG.METRICS])
                rs.append(np.array(per_user_reward, dtype=np.float64))
                ps.append(_pw_arr.astype(np.float64))
            record_reward /= n_step                                   # TRAIN:1824
            out = [float(np.mean(raw_global[-n_step:])),              # TRAIN:1838-1839 (learn reward = env reward, :1614)
                   ep_sum["last_off_kbit_sum"], ep_sum["last_local_kbit_sum"],
                   float(getattr(env, "last_mec_queue_cycles", 0.0)),                      # TRAIN:1974
                   ep_sum["last_backlog_kbit_mean"] / ep_steps, ep_sum["last_delay_local_mean"] / ep_steps,
                   ep_sum["last_delay_edge_q_mean"] / ep_steps, ep_sum["last_delay_edge_c_mean"] / ep_steps,
                   ep_sum["last_t_tx_mean"] / ep_steps, ep_sum["last_mec_utilization"] / ep_steps,
                   ep_sum["last_local_util_mean"] / ep_steps, ep_sum["last_qos_violation"] / ep_steps,
                   ep_sum["last_delay_mean"] / ep_steps, ep_sum["last_energy_mean"] / ep_steps,
                   float(np.mean(Power_offload)), float(np.mean(Power_local)), float(np.mean(Power)),   # TRAIN:1841-1843
                   float(np.min(record_reward)), float(np.var(record_reward)), jain(record_reward),   # TRAIN:1939-1941
                   ep_env_best]
            m_env.append(ms); r_env.append(rs); p_env.append(ps); ep_env.append(out)
        rec["metrics"].append(m_env); rec["reward"].append(r_env); rec["power_w"].append(p_env); rec["episode"].append(ep_env)
    d = {k: np.array(v) for k, v in rec.items()}          # metrics [env, ep, step, 14] ... episode [env, ep, 21]
    d["shape"] = np.array([V, n_env, n_ep, n_step])
    rng = np.random.default_rng(seed)
    xs = [rng.normal(-1.0, 1.5, V) for _ in range(6)] + [np.zeros(V), np.full(V, -2.5), np.array([1e-9] + [0.0] * (V - 1))]
    d["jain_x"] = np.array(xs)
    d["jain_y"] = np.array([jain(x) for x in xs])
    return d


if __name__ == "__main__":
    os.makedirs(CG.OUT_DIR, exist_ok=True)
    CG.save("episode_metrics_8.npz", capture(8, 6, 2, 25, 1300))
    CG.save("episode_metrics_5.npz", capture(5, 3, 2, 12, 1301))
