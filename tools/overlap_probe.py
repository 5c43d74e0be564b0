#!/usr/bin/env python3
"""Timing probe (results are NOT meaningful: the candidate indices are raced on purpose): the indexed BCD sweep on a
second stream beside the fused gains+step kernel, vs the two back to back on one stream.  Round 3: theta kept by index
(the sweep writes indices only, the step reads them).  Usage: overlap_probe.py [priority]  (priority = 1: the sweep's
stream gets the high priority)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import build_env, synthetic_groups
E, V, M = 32768, 16, 256
dev = torch.device("cuda:0")
env = build_env(E, V, M, dev, 0, 0)
env.lazy_theta = True
rng = np.random.default_rng(0)
action = torch.from_numpy(rng.uniform(0, 1, (E, 2, V)).astype(np.float32)).to(dev)
p, n = synthetic_groups(E, V, rng)
partner, ng = torch.from_numpy(p).to(dev), torch.from_numpy(n).to(dev)
step = env.bind_step(action, partner, ng, None, fused=True)
env.optimize_phase_shift(); env.optimize_phase_shift()
s2 = torch.cuda.Stream(priority=-1) if len(sys.argv) > 1 and sys.argv[1] == "1" else torch.cuda.Stream()

def serial(n):
    for _ in range(n):
        env.optimize_phase_shift(); step()

def overlapped(n):
    for _ in range(n):
        with torch.cuda.stream(s2):
            env.optimize_phase_shift()
        step()
        torch.cuda.current_stream().wait_stream(s2)
        s2.wait_stream(torch.cuda.current_stream())

def only_step(n):
    for _ in range(n):
        step()

def only_sweep(n):
    for _ in range(n):
        env.optimize_phase_shift()

out = {}
for name, fn in (("serial", serial), ("overlapped", overlapped), ("only_step", only_step), ("only_sweep", only_sweep)):
    fn(20); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(200); torch.cuda.synchronize()
    out[name + "_us"] = (time.perf_counter() - t0) / 200 * 1e6
print(json.dumps(out))
