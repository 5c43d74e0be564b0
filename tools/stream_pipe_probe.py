#!/usr/bin/env python3
"""Probe: what a second stream costs.  The NOMA bookkeeping launch sits between two step launches today (step -> group -> step:
33.6 us per step against 26.3 without it).  If the group launch ran on a SECOND stream, concurrently with the step it
does not feed, the critical path would be the step kernel plus the cross-stream event waits -- this measures that skeleton:
  a) step only;  b) step, group on one stream (today);  c) step on stream A, group on stream B, each waiting for the
  other's PREVIOUS launch (events)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bench import build_env
from ris_vec_marl_amd import NomaGrouper

E, V, M = 32768, 8, 64
dev = torch.device("cuda:0")
env = build_env(E, V, M, dev, 0, 0)
rng = np.random.default_rng(0)
action = torch.from_numpy(rng.uniform(0, 1, (E, 2, V)).astype(np.float32)).to(dev)
g = NomaGrouper(env)
env.update_channel_gains()
g.begin_episode(0); g.refresh_mask()
partner, n_groups = g.group(action[:, 0, :].contiguous(), 0)
step = env.bind_step(action, partner, n_groups, None, fused=True, metrics=True, power_w=False, obs=True)
group = g.bind_group(action[:, 0, :].contiguous())
step(); group(1)
torch.cuda.synchronize()


def timed(fn, n=400):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(20):
        fn(i)
    torch.cuda.synchronize()
    a.record()
    for i in range(n):
        fn(20 + i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


sA = torch.cuda.current_stream(dev)
sB = torch.cuda.Stream(device=dev)
evA = [torch.cuda.Event() for _ in range(4)]
evB = [torch.cuda.Event() for _ in range(4)]
for e in evA + evB:
    e.record(sA)


def one(i):
    step()


def two(i):
    group(2 + i); step()


def piped(i):
    k = i % 4
    sA.wait_event(evB[(i - 1) % 4])          # step(i) needs the side launch of the previous step
    step()
    evA[k].record(sA)
    with torch.cuda.stream(sB):
        sB.wait_event(evA[(i - 1) % 4])      # side launch (i) needs step (i-1)
        group(2 + i)
        evB[k].record(sB)


out = dict(step_only_us=round(timed(one), 2), step_group_one_stream_us=round(timed(two), 2),
           step_and_group_two_streams_us=round(timed(piped), 2))
torch.cuda.synchronize()
print(json.dumps(out))
