#!/usr/bin/env python3
"""The reference driver's rollout loop (marl_train_bcd.py:1244-1830, without the learner) with every
stage on the GPU: batched SAC policies -> NOMA pairing -> fused RIS gains + step() -> replay ring ->
per-episode metrics (written as a TensorBoard event file under runs/rollout).

    python examples/rollout.py [n_envs] [episodes] [log_dir]

Everything between the policy weights and the sampled training batch stays in HBM; the only host work
per step is a handful of kernel launches.  Needs an MI355X and the built librisvec.so."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ris_vec_marl_amd import (BatchedPolicy, EpisodeMeter, NomaGrouper, ScalarSink, VecEnviron, VecReplayBuffer,  # noqa: E402
                              apply_yaml_config, reference_lanes)

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
EPISODES = int(sys.argv[2]) if len(sys.argv) > 2 else 3
LOG_DIR = sys.argv[3] if len(sys.argv) > 3 else os.path.join("runs", "rollout")
V, M, N_STEP, RIS_EVERY = 8, 40, 100, 100            # Config defaults of the driver (n_veh, M, steps, K_STEPS)

L = reference_lanes()
env = VecEnviron(L["down_lanes"], L["up_lanes"], L["left_lanes"], L["right_lanes"], 400, 400, V, M, 3,
                 n_envs=E, device="cuda:0", seed=0)
apply_yaml_config(env, None)                        # or load_yaml("config.yaml")
env.make_new_game()
policy = BatchedPolicy(V, 5, 512, 256, device="cuda:0")          # policy.load_agent_state_dict(i, sd) for trained weights
grouper = NomaGrouper(env)
memory = VecReplayBuffer(8 * N_STEP * E, 5, V + 2, V, device="cuda:0")
meter, writer = EpisodeMeter(env), ScalarSink(LOG_DIR)           # the driver's ep_* sums and its SummaryWriter

# Tensors every stage reads / writes in place, and one pre-marshalled launcher per stage: inside the step loop
# the host does nothing but issue launches (the unbound methods -- env.step(...), memory.store_batch(...) --
# do the same work with the argument checking on every call).
action_env = torch.zeros(E, 2, V, device="cuda:0")               # env action            (TRAIN:1601-1608)
p_off01 = torch.zeros(E, V, device="cuda:0")                     # pairing power         (TRAIN:1391-1396)
action_store = torch.zeros(E, V * (V + 2), device="cuda:0")      # replay action row     (TRAIN:1776-1784)
env.update_channel_gains()
grouper.begin_episode(0)
mask = grouper.refresh_mask()                                    # [E,V,V] uint8, updated in place by later refreshes
partner, n_groups = grouper.group(p_off01, 0)                    # state views the step launcher can bind
step = env.bind_step(action_env, partner, n_groups)
group = grouper.bind_group(p_off01)
store = memory.bind_store(None, action_store, env.tensors["metrics"], env.tensors["reward"], env.tensors["obs"], mask)
count = meter.bind(env)
policy.forward_heads(env.tensors["obs"])                         # untimed: first-launch costs (code objects, weight preparation)

torch.cuda.synchronize()
t0 = time.perf_counter()
for ep in range(EPISODES):
    env.begin_episode(ep, env_refresh_every=5)
    grouper.begin_episode(ep)
    meter.begin_episode()
    for st in range(N_STEP):
        refreshed = env.begin_step(st, ris_every=RIS_EVERY)       # BCD sweep + gains on refresh steps
        if refreshed:
            grouper.refresh_mask()
        policy.choose_action(env.tensors["obs"], mask, cpu_share_floor=env.cpu_share_floor, want_onehot=False,
                             out=(action_env, p_off01, action_store))
        group(st)
        step()
        store(done=(st == N_STEP - 1), use_mask=refreshed)
        count()
    batch = memory.sample_buffer(256)               # what global_learn would consume
    sc = writer.write_episode(meter, ep)            # the driver's tags, mean over the envs
    print("episode %d: global reward %.3f, delay %.2f ms, energy %.2e J, QoS violations %.1f %%, Jain %.3f, "
          "pairs/env %.2f, replay rows %d"
          % (ep, sc["reward/global_avg"], sc["abs/delay_ms"], sc["abs/energy_J"], 100 * sc["qos/violation_rate_ep_mean"],
             sc["reward/jain"], float(V - grouper._t["n_groups"].float().mean()), min(memory.mem_cntr, memory.mem_size)))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
writer.close()
print("%.2e env-steps/s over %d envs x %d steps; scalars in %s" % (E * EPISODES * N_STEP / dt, E, EPISODES * N_STEP, writer.path))
