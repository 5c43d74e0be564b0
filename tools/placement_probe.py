#!/usr/bin/env python3
"""Probe: is the two-level effect on streams beyond the Infinity Cache (EXPERIMENTS.md, round 3) a property of WHERE the
h_r allocation landed?  One process, the same env, h_r re-allocated several times (clone + pointer swap): step time per
allocation.  Usage: placement_probe.py [E V M]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import build_env, synthetic_groups
E, V, M = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (262144, 8, 64)))
dev = torch.device("cuda:0")
env = build_env(E, V, M, dev, 0, 0)
rng = np.random.default_rng(0)
action = torch.from_numpy(rng.uniform(0, 1, (E, 2, V)).astype(np.float32)).to(dev)
p, n = synthetic_groups(E, V, rng)
partner, ng = torch.from_numpy(p).to(dev), torch.from_numpy(n).to(dev)
step = env.bind_step(action, partner, ng, None, fused=True)

def timed(k=100):
    for _ in range(10):
        step()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k):
        step()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / k

def read_us(t, k=5):
    """a plain streaming read of the allocation (torch reduction), best of k"""
    best = 1e30
    for _ in range(k):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); t.sum(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3)
    return round(best, 1)

out = [dict(alloc=0, ptr=hex(env._t["h_r"].data_ptr()), us=round(timed(), 2), read_us=read_us(env._t["h_r"]))]
keep = []
for i in range(1, 6):
    new = env._t["h_r"].clone()
    keep.append(env._t["h_r"])          # keep the old block alive so that the allocator hands out a NEW one
    env._t["h_r"] = new
    env._cstate.h_r = new.data_ptr()
    out.append(dict(alloc=i, ptr=hex(new.data_ptr()), us=round(timed(), 2), read_us=read_us(new)))
out.append(dict(alloc="first again", us=round((lambda: (setattr(env._cstate, "h_r", keep[0].data_ptr()), timed())[1])(), 2)))
print(json.dumps(out))
