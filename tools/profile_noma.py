#!/usr/bin/env python3
"""Time the NOMA grouping kernels (f2) at one shape: mask rebuild, group() when every env re-solves
its pairing, group() when every env is frozen.  HIP-event timing on the launch stream; prints one
JSON line per case.   usage: profile_noma.py [E V reps]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ris_vec_marl_amd import NomaGrouper, VecEnviron, reference_lanes, apply_yaml_config  # noqa: E402

E, V, reps = (int(x) for x in (sys.argv[1:4] + ["32768", "8", "20"][len(sys.argv) - 1:]))
M = 64
L = reference_lanes()
env = VecEnviron(L["down_lanes"], L["up_lanes"], L["left_lanes"], L["right_lanes"], 400, 400, V, M, 3,
                 n_envs=E, device="cuda:0", seed=0)
apply_yaml_config(env, None)
env.make_new_game(); env.renew_positions(); env.compute_parms(); env.Random_phase(); env.update_channel_gains()
rng = np.random.default_rng(0)
p01 = torch.from_numpy(rng.uniform(0, 1, (E, V)).astype(np.float32)).cuda()
reward = torch.from_numpy((-rng.uniform(1, 5, E)).astype(np.float32)).cuda()


def timed(fn, n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


CASES = [("driver-default", False), ("config.yaml", True)]
if os.environ.get("NOMA_PROFILE_NO_QOS"):
    CASES.append(("config.yaml without the QoS check", "noqos"))
for name, yaml in CASES:
    g = NomaGrouper(env)
    if yaml:
        g.config.apply_yaml({"min_pair_target": 3, "mwm_backoff_rounds": 3, "qos_enable": yaml != "noqos",
                             "reward": {"mask_topk_start": 7, "mask_topk_end": 7, "mask_tau_q_start": 0.10,
                                        "mask_tau_q_end": 0.25, "pairing_threshold_quantile": 0.25}})
    g.begin_episode(0)
    t_mask = timed(lambda: g.refresh_mask(), reps)

    def solve():
        g.begin_episode(0)          # forget the frozen groups: every env re-solves
        g.refresh_mask()
        g.group(p01, 0)
    t_all = timed(solve, reps)
    t_begin = timed(lambda: g.begin_episode(0), reps)
    g.begin_episode(0); g.refresh_mask(); g.group(p01, 0); g.group(p01, 1, prev_global=reward)
    info = g.info.cpu().numpy()
    t_frozen = timed(lambda: g.group(p01, 2, prev_global=reward), reps)
    bound = g.bind_group(p01)
    t_frozen_bound = timed(lambda: bound(3), 200)
    g.begin_episode(0); g.refresh_mask(); g.group(p01, 0)
    info0 = g.info.cpu().numpy()
    t_solve = t_all - t_mask - t_begin
    print(json.dumps(dict(config=name, E=E, V=V, mask_us=round(t_mask, 1), solve_us=round(t_solve, 1),
                          frozen_us=round(t_frozen, 1), frozen_bound_us=round(t_frozen_bound, 1), solves_per_s=round(E / (t_solve * 1e-6)),
                          mean_pairs=float(info0[:, 2].mean()), backoff_frac=float((info0[:, 1] > 0).mean()),
                          mean_matchable=float(info0[:, 3].mean()))))
