#!/usr/bin/env python3
"""Launch every kernel of the library N times at one shape so that
`rocprofv3 --kernel-trace --stats -- python3 tools/profile_all_kernels.py` yields a per-kernel
average duration; prints the algorithmic bytes each launch must move, for the roofline table in
DESIGN.md.   usage: profile_all_kernels.py [E V M reps]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ris_vec_marl_amd import (VecEnviron, reference_lanes, apply_yaml_config, NomaGrouper, VecReplayBuffer,  # noqa: E402
                              marshal_actions, BatchedPolicy, EpisodeMeter)

E, V, M, reps = (int(x) for x in (sys.argv[1:5] + ["32768", "8", "64", "20"][len(sys.argv) - 1:]))
L = reference_lanes()
env = VecEnviron(L["down_lanes"], L["up_lanes"], L["left_lanes"], L["right_lanes"], 400, 400, V, M, 3,
                 n_envs=E, device="cuda:0", seed=0)
apply_yaml_config(env, None)
rng = np.random.default_rng(0)
action = torch.from_numpy(rng.uniform(0, 1, (E, 2, V)).astype(np.float32)).cuda()
partner = torch.full((E, V), -1, dtype=torch.int32).cuda()
ng = torch.full((E,), V, dtype=torch.int32).cuda()
phase = torch.from_numpy(rng.uniform(0, 6.28, (E, M)).astype(np.float32)).cuda()
pw = action[:, 0, :].contiguous()
actions_T = torch.from_numpy(rng.uniform(0, 1, (16, E, 2, V)).astype(np.float32)).cuda()
grouper = NomaGrouper(env) if V <= 16 else None
replay = VecReplayBuffer(3 * E, 5, V + 2, V, device="cuda:0")
power_raw = torch.from_numpy(rng.uniform(-1, 1, (E, V, 2)).astype(np.float32)).cuda()
probs = torch.from_numpy(rng.dirichlet(np.ones(V), (E, V)).astype(np.float32)).cuda()
policy = BatchedPolicy(V, 5, 512, 256, device="cuda:0")                       # k_policy_mlp (one launch) where built for the shape
policy3 = BatchedPolicy(V, 5, 512, 256, device="cuda:0", gemm="fp16x3")       # the three-launch form
meter = EpisodeMeter(env)
for _ in range(reps):
    env.make_new_game()
    env.renew_positions()
    env.compute_parms()                # k_geometry + k_colsum
    env.Random_phase()
    env.get_next_phase(phase)
    env.optimize_phase_shift()         # k_bcd_sweep: the generic sweep (theta set by hand: indices unknown), column sums cached
    env.optimize_phase_shift()         # k_bcd_sweep8_pair: the indexed sweep (indices left by the previous one, cached sum)
    env.update_channel_gains()         # k_gain
    env.data_rate(pw, partner, ng)
    env.step(action, partner, ng, None, fused=False, power_w=False)
    env.step(action, partner, ng, None, fused=True, power_w=False)
    env.step(action, partner, ng, None, fused=True, power_w=False, steer=True)      # k_step_steer
    if (V, M) in ((8, 64), (8, 36), (8, 40), (4, 16)):
        env.step_many(actions_T, partner, ng, None)                   # k_step_fused_lat<.., MULTI>: T = 16 steps in one launch
    env.step_many(actions_T, partner, ng, None, fused=False)          # k_step_multi: T = 16 steps on the cached gains, any shape
    env.sarl_step(action, phase)
    env.channel_model = "3gpp_umi"
    env.update_channel_gains()
    env.channel_model = "free"
    env.update_channel_gains()
    a_env, p01, a_store = marshal_actions(power_raw, probs, 0.1)      # k_marshal_actions
    mask = None
    if grouper is not None:
        grouper.begin_episode(0)
        mask = grouper.refresh_mask()                                   # k_noma_mask
        grouper.group(p01, 0)                                          # k_noma_group (all envs solve)
        grouper.group(p01, 2)                                          # frozen step
        grouper.flush()                                                # k_noma_flush
    obs = env.tensors["obs"]
    replay.store_batch(obs, a_store, env.tensors["metrics"], env.tensors["reward"], obs, False, mask)   # k_replay_store
    replay.sample_buffer(4096)                                         # k_replay_sample
    policy.choose_action(obs, mask, cpu_share_floor=0.1)               # k_policy_mlp + k_policy_sample
    policy3.forward_heads(obs)                                         # k_policy_layer1 (split fp16) + fp16 GEMM + k_policy_heads
    meter.begin_episode()                                              # k_episode_clear
    meter.accumulate(env)                                              # k_episode_accumulate
    meter.summarize()                                                  # k_episode_summary + k_episode_fold
torch.cuda.synchronize()
B = dict(
    k_reset=E * V * (16 + 4 + 4 + 4), k_mobility=E * V * (16 + 4 + 4 + 16 + 4), k_geometry=E * V * (16 + 12 + 8 * M),
    k_colsum=E * (8 * V * M + 16 * M), k_random_phase=E * 8 * M, k_set_phase=E * 12 * M,
    k_bcd_sweep=E * (2 * 16 * M + 2 * 8 * M + 8 * M), k_bcd_sweep8_pair=E * (16 * M + 2 * M + 8 * M + 32), k_theta_from_index=E * 9 * M,
    k_colsum_slab=E * (8 * V * M + 16 * M),
    k_step_fused_lat=E * (8 * V * M + 8 * M + 16 * (8 * V + 24 * V + 64) + 40 * V + 8), k_gain=E * (8 * V * M + 8 * M + 8 * V),
    k_data_rate=E * 16 * V + 4 * E, k_step=E * (60 * V + 68), k_step_multi=E * 16 * (8 * V + 24 * V + 64) + E * (60 * V + 68), k_step_fused=E * (8 * V * M + 8 * M + 64 * V + 68),
    k_step_steer=E * (16 * V + 8 * M + 64 * V + 68),
    k_sarl_step=E * (8 * V * M + 8 * M + 48 * V + 4), k_gain_3gpp=E * V * 20,
    # f2 / f3 (float words read + written; the NOMA kernels are latency-bound, bytes listed for completeness)
    k_replay_store=E * (2 * 4 * (2 * 5 * V + V * (V + 2) + V + 1) + 5 * V * V + 1),
    k_replay_sample=4096 * (2 * 4 * (2 * 5 * V + V * (V + 2) + V + 1 + V * V) + 2 + 8),
    k_marshal_actions=E * (V * (8 + 4 * V) + 12 * V + 4 * V * (V + 2)),
    k_marshal_pairs=E * (V * (8 + 4 * V) + 12 * V + 4 * V * (V + 2)),
    k_noma_mask=E * (4 * V + V * V + 8), k_noma_group=E * (8 * V + 4 * V * V * 2 + V * V + 16 * V + 40),
    k_noma_flush=E * (8 * V * V + 8 * V + 4),
    # batched policy (5-512-256): layer1 writes h1, heads reads the fc2 product; sample: heads + mask in, all outputs out
    k_policy_layer1=E * V * (20 + 3 * 2 * 512), k_policy_heads=E * V * (4 * 256 + 4 * (4 + V)),
    k_policy_mlp=E * V * (20 + 4 * (4 + V)),          # HBM bytes; the kernel is matrix-core bound (see "flops" in kernel_table)
    k_policy_sample=E * V * (4 * (4 + V) + V + 8 + 4 * V + 4 * V + 12 + 4 * (V + 2)),
    # f4 episode meter: float64 accumulators read + written, the step's metrics / reward / power_w read
    k_episode_clear=E * 8 * (17 + V), k_episode_accumulate=E * (16 * (17 + V) + 56 + 12 * V),
    k_episode_summary=E * (8 * (17 + V) + 4 + 8 * 21))
print(json.dumps(dict(E=E, V=V, M=M, reps=reps, algorithmic_bytes=B)))
