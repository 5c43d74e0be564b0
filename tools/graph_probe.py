#!/usr/bin/env python3
"""Does a hipGraph shorten the boundary between dependent step launches?  (asked for the
rollout loop in a hipGraph; DESIGN.md answered from the MI355X price list -- this measures it.)

For each shape: the SAME 32 fused-step launches (one env batch, each step depending on the one before) issued
(a) eagerly through the pre-bound C-ABI launcher and (b) as one captured graph replayed; µs per step from HIP events
over 100 replays.  The captured launches carry the Philox counters of the capture (a graph replays its arguments), so
(b) is a timing probe only, not a product path -- that would need the counter in device memory first.
Usage: python tools/graph_probe.py  -> one JSON line per shape."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import build_env, synthetic_groups

T, REPS = 32, 100
dev = torch.device("cuda:0")
side = torch.cuda.Stream(device=dev)


def probe(E, V, M, fused=True):
    rng = np.random.default_rng(0)
    env = build_env(E, V, M, dev, 0, 0)
    action = torch.from_numpy(rng.uniform(0, 1, (E, 2, V)).astype(np.float32)).to(dev)
    p, n = synthetic_groups(E, V, rng)
    partner, ng = torch.from_numpy(p).to(dev), torch.from_numpy(n).to(dev)
    if not fused:
        env.update_channel_gains()
    torch.cuda.synchronize()
    out = {}
    with torch.cuda.stream(side):
        step = env.bind_step(action, partner, ng, None, fused=fused)      # bound to `side`: the stream a capture records
        for _ in range(3 * T):
            step()
        side.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(side)
        for _ in range(REPS * T):
            step()
        b.record(side)
        side.synchronize()
        out["eager_us"] = a.elapsed_time(b) * 1e3 / (REPS * T)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(T):
            step()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(REPS):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    out["graph_us"] = a.elapsed_time(b) * 1e3 / (REPS * T)
    return out


if __name__ == "__main__":
    for name, E, V, M, fused in (("configs[1] 4096x8x36 fused", 4096, 8, 36, True),
                                 ("configs[3] shard 8192x8x64 fused", 8192, 8, 64, True),
                                 ("32768x8x64 cached step", 32768, 8, 64, False),
                                 ("configs[2] 32768x8x64 fused", 32768, 8, 64, True)):
        r = probe(E, V, M, fused)
        print(json.dumps({"shape": name, "steps_per_graph": T, "eager_us_per_step": round(r["eager_us"], 3),
                          "graph_us_per_step": round(r["graph_us"], 3)}), flush=True)
