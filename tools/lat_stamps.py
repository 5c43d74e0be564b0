#!/usr/bin/env python3
"""Diagnostic: per-wavefront timeline of the latency-shaped fused step at BASELINE configs[1] (4 096 x 8 x 36):
s_memrealtime (100 MHz) at entry / loads issued / cascade reduced / step() done / stores drained."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the stamp build lives in the DIAGNOSTIC library only (make -C ris_vec_marl_amd/csrc diag)
os.environ["RISVEC_LIB"] = os.path.join(ROOT, "ris_vec_marl_amd", "csrc", "librisvec_diag.so")
sys.path.insert(0, ROOT)
import numpy as np, torch
from bench import build_env, synthetic_groups
E, V, M = 4096, 8, 36
dev = torch.device("cuda:0")
env = build_env(E, V, M, dev, 0, 0)
rng = np.random.default_rng(0)
action = torch.from_numpy(rng.uniform(0, 1, (E, 2, V)).astype(np.float32)).to(dev)
p, n = synthetic_groups(E, V, rng)
partner, ng = torch.from_numpy(p).to(dev), torch.from_numpy(n).to(dev)
step = env.bind_step(action, partner, ng, None, fused=True)
for _ in range(50):
    step()
dbg = torch.zeros(E // 2 * 5, dtype=torch.int64, device=dev)
os.environ["RISVEC_LAT_STAMPS_PTR"] = str(dbg.data_ptr())
res = []
for _ in range(20):
    step(); step(); step()
    torch.cuda.synchronize()
    t = dbg.cpu().numpy().reshape(-1, 5).astype(np.float64) * 10.0          # ns
    t0 = t[:, 0].min()
    res.append(dict(first_entry=0.0, last_entry=float(t[:, 0].max() - t0), median_entry=float(np.median(t[:, 0]) - t0),
                    issue=float(np.median(t[:, 1] - t[:, 0])), loads=float(np.median(t[:, 2] - t[:, 1])),
                    step=float(np.median(t[:, 3] - t[:, 2])), drain=float(np.median(t[:, 4] - t[:, 3])),
                    last_exit=float(t[:, 4].max() - t0), p90_wave_life=float(np.percentile(t[:, 4] - t[:, 0], 90))))
keys = res[0].keys()
print(json.dumps({k: round(float(np.median([r[k] for r in res])), 1) for k in keys}))
