#!/usr/bin/env python3
"""Soak run on one GPU: every launch form of the hot path for a few thousand steps each, outputs checked for NaN / Inf
and for the invariants that do not need an oracle (queues >= 0, rewards inside the clip, Philox stream identical between
a T-step launch and T single launches at the end of a long run).  Prints one JSON line per phase.
Usage: python tools/soak.py [scale]   (scale 1.0 = about a minute on an MI355X)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import build_env, synthetic_groups

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
dev = torch.device("cuda:0")


def finite(env, keys=("gain", "reward", "data_buf", "mec_q", "rate", "obs", "metrics", "theta")):
    for k in keys:
        t = env.tensors[k]
        assert torch.isfinite(t).all(), k
    assert (env.tensors["data_buf"] >= 0).all() and (env.tensors["mec_q"] >= 0).all()
    clip = float(env.params.reward_clip)
    assert (env.tensors["reward"].abs() <= clip * (1 + 1e-6)).all()


def phase(name, E, V, M, n, lazy=False, **kw):
    rng = np.random.default_rng(1)
    env = build_env(E, V, M, dev, 3, 0)
    env.lazy_theta = lazy
    action = torch.from_numpy(rng.uniform(0, 1, (E, 2, V)).astype(np.float32)).to(dev)
    p, g = synthetic_groups(E, V, rng)
    partner, ng = torch.from_numpy(p).to(dev), torch.from_numpy(g).to(dev)
    kw.setdefault("fused", True)
    if not kw["fused"]:
        env.update_channel_gains()
    step = env.bind_step(action, partner, ng, None, **kw)
    t0 = time.perf_counter()
    for i in range(n):
        step()
        if i % 500 == 499:
            finite(env)
    torch.cuda.synchronize()
    finite(env)
    print(json.dumps({"phase": name, "steps": n, "us_per_step": round((time.perf_counter() - t0) / n * 1e6, 2)}), flush=True)
    return env


def tstep_phase(name, E, V, M, T, launches, fused):
    rng = np.random.default_rng(2)
    a = build_env(E, V, M, dev, 5, 0)
    b = build_env(E, V, M, dev, 5, 0)
    p, g = synthetic_groups(E, V, rng)
    partner, ng = torch.from_numpy(p).to(dev), torch.from_numpy(g).to(dev)
    actions = torch.from_numpy(rng.uniform(0, 1, (T, E, 2, V)).astype(np.float32)).to(dev)
    if not fused:
        a.update_channel_gains(); b.update_channel_gains()
    many = a.bind_step_many(actions, partner, ng, None, fused=fused)
    for _ in range(launches):
        many()
    singles = [b.bind_step(actions[t], partner, ng, None, fused=fused, power_w=False) for t in range(T)]
    for _ in range(launches):
        for s in singles:
            s()
    torch.cuda.synchronize()
    for k in ("data_buf", "mec_q", "reward", "obs", "metrics", "rate"):
        assert torch.equal(a.tensors[k], b.tensors[k]), (name, k)
    finite(a)
    print(json.dumps({"phase": name, "steps": T * launches, "identical_to_single_launches": True}), flush=True)


def noma_phase(name, E, V, M, episodes, steps=100):
    """Rollout with the device pairing stage: a full re-solve at the start of every episode, frozen steps in between, random
    in-episode repairs (unstick draws from the device Philox stream); groups checked for consistency every episode."""
    from ris_vec_marl_amd import NomaGrouper
    rng = np.random.default_rng(4)
    env = build_env(E, V, M, dev, 7, 0)
    action = torch.from_numpy(rng.uniform(0, 1, (E, 2, V)).astype(np.float32)).to(dev)
    p01 = action[:, 0, :].contiguous()
    gr = NomaGrouper(env)
    gr.config.freeze_unstick_prob = 0.002
    gr.config.qos_enable = True
    env.update_channel_gains()
    gr.begin_episode(0); gr.refresh_mask()
    partner, ng = gr.group(p01, 0)
    step = env.bind_step(action, partner, ng, None, fused=True)
    group = gr.bind_group(p01)
    solved = 0
    t0 = time.perf_counter()
    for ep in range(episodes):
        gr.begin_episode(ep); gr.refresh_mask()
        gr.group(p01, 0)
        for t in range(1, steps):
            step()
            group(t)
        solved += int((gr._t["pending"] < steps - 1).sum().item())     # envs that re-solved inside the episode
        step()
        part = partner.cpu().numpy()
        busy = part >= 0
        mate = np.where(busy, part & 0xFFFF, 0)
        assert np.array_equal(np.take_along_axis(mate, mate, 1)[busy], np.broadcast_to(np.arange(V), part.shape)[busy]), name
        assert np.array_equal(ng.cpu().numpy(), V - busy.sum(1) // 2), name
        if V > 8:
            assert int(gr._t["scratch"][:4].view(torch.int32)[0].item()) == 0, name     # the second launch's list is empty
        finite(env)
    torch.cuda.synchronize()
    assert solved > 0                                   # in-episode repairs happened
    print(json.dumps({"phase": name, "steps": episodes * steps, "envs_that_resolved_inside_an_episode": solved,
                      "us_per_step": round((time.perf_counter() - t0) / (episodes * steps) * 1e6, 2)}), flush=True)


n = lambda x: max(10, int(x * scale))
phase("headline fused 32768x8x64", 32768, 8, 64, n(4000))
phase("configs[1] fused 4096x8x36", 4096, 8, 36, n(8000))
phase("configs[3] shard fused 8192x8x64", 8192, 8, 64, n(8000))
phase("configs[4] bcd 32768x16x256", 32768, 16, 256, n(600), bcd=True)
phase("configs[4] bcd, theta by index", 32768, 16, 256, n(600), lazy=True, bcd=True)
phase("bcd, theta by index, run-time M 8192x8x100", 8192, 8, 100, n(1500), lazy=True, bcd=True)
phase("beyond the cache 262144x8x64", 262144, 8, 64, n(300))
phase("just beyond the cache (alternating walk) 65536x8x64", 65536, 8, 64, n(1500))
phase("run-time M 32768x8x120", 32768, 8, 120, n(2000))
phase("run-time M 20000x16x50", 20000, 16, 50, n(2000))
phase("run-time M 50000x4x24", 50000, 4, 24, n(4000))
phase("cached step 32768x8", 32768, 8, 64, n(8000), fused=False)
phase("reference default 16384x8x40", 16384, 8, 40, n(4000))
tstep_phase("T-step fused 4096x8x36", 4096, 8, 36, 32, n(60), True)
tstep_phase("T-step cached 2000x16x256", 2000, 16, 256, 25, n(40), False)
tstep_phase("T-step fused fallback 700x6x50", 700, 6, 50, 8, n(30), True)
noma_phase("pairing stage in the loop 32768x8x64", 32768, 8, 64, max(2, int(12 * scale)))
noma_phase("pairing stage in the loop 8192x16x64 (two launches)", 8192, 16, 64, max(2, int(12 * scale)))
noma_phase("pairing stage in the loop 1000x11x40", 1000, 11, 40, max(2, int(12 * scale)))
print(json.dumps({"soak": "ok"}))
