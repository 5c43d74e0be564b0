#!/usr/bin/env python3
"""Diagnostic: where the time of a 16-vehicle NOMA re-solve goes (the s_memtime build of k_noma_group in the DIAGNOSTIC
library, `make -C ris_vec_marl_amd/csrc diag`): ticks per phase, summed over each wavefront's solves, median over
wavefronts.  Usage: noma_stamps.py [E]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["RISVEC_NOMA_STAMPS"] = "1"
os.environ["RISVEC_LIB"] = os.path.join(ROOT, "ris_vec_marl_amd", "csrc", "librisvec_diag.so")
sys.path.insert(0, ROOT)
import numpy as np, torch
from ris_vec_marl_amd import NomaGrouper, VecEnviron, reference_lanes, apply_yaml_config

E = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
V, M = 16, 64
L = reference_lanes()
env = VecEnviron(L["down_lanes"], L["up_lanes"], L["left_lanes"], L["right_lanes"], 400, 400, V, M, 3, n_envs=E,
                 device="cuda:0", seed=0)
apply_yaml_config(env, None)
env.make_new_game(); env.renew_positions(); env.compute_parms(); env.Random_phase(); env.update_channel_gains()
p01 = torch.from_numpy(np.random.default_rng(0).uniform(0, 1, (E, V)).astype(np.float32)).cuda()
NAMES = ["loads+replay+qos", "scores+ranks", "threshold+weights", "frontiers", "table", "walk", "completion", "relaxation",
         "stores", "dense_fallback", "n_frontier_tables", "n_solves"]
for name, yaml in (("driver-default", False), ("config.yaml", True)):
    g = NomaGrouper(env)
    if yaml:
        g.config.apply_yaml({"min_pair_target": 3, "mwm_backoff_rounds": 3, "qos_enable": True,
                             "reward": {"mask_topk_start": 7, "mask_topk_end": 7, "mask_tau_q_start": 0.10,
                                        "mask_tau_q_end": 0.25, "pairing_threshold_quantile": 0.25}})
    for _ in range(2):
        g.begin_episode(0); g.refresh_mask(); g.group(p01, 0)
    torch.cuda.synchronize()
    n_stamp = (512 * 64 + 512) * 12 * 8
    allb = g._t["scratch"].cpu().numpy()[-n_stamp:].copy().view(np.int64).reshape(-1, 12)
    raw = allb[: (E + 7) // 8]                  # first launch, one row per block of 8 envs
    second = allb[512 * 64:]
    second = second[second[:, 11] > 0]
    med = np.median(raw, axis=0)
    tot = float(med[:10].sum())
    out = {n: round(float(v), 1) for n, v in zip(NAMES, med)}
    out["share"] = {n: round(float(v) / tot, 3) for n, v in zip(NAMES[:10], med[:10])}
    print(json.dumps(dict(config=name, E=E, V=V, blocks=int(raw.shape[0]), second_launch_envs=int(second[:, 11].sum()),
                          second_launch_ticks_per_solve=float(np.median(second[:, :10].sum(axis=1) / second[:, 11])) if len(second) else 0.0,
                          max_block_ticks=float(raw[:, :10].sum(axis=1).max()),
                          block_ticks_mean=float(raw[:, :10].sum(axis=1).mean()),
                          block_ticks_percentiles_10_50_90_99=[float(x) for x in np.percentile(raw[:, :10].sum(axis=1), [10, 50, 90, 99])],
                          ticks_per_block=tot,
                          ticks_per_solve=tot / max(1.0, float(med[11])), **out)))
