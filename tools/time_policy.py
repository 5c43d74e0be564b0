#!/usr/bin/env python3
"""Time BatchedPolicy.forward_heads (HIP events, device-resident inputs) for each forward mode.
    python tools/time_policy.py [n_envs] [n_agents] [fc1] [fc2]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ris_vec_marl_amd import BatchedPolicy  # noqa: E402

E, V, F1, F2 = (int(x) for x in (sys.argv[1:5] + ["32768", "8", "512", "256"][len(sys.argv) - 1:]))
obs = torch.rand(E, V, 5, device="cuda:0") * 1.2
out = {"E": E, "V": V, "fc1": F1, "fc2": F2}
for mode in BatchedPolicy.GEMM_MODES:
    try:
        pol = BatchedPolicy(V, 5, F1, F2, device="cuda:0", gemm=mode)
    except ValueError:
        continue
    for _ in range(5):
        pol.forward_heads(obs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        pol.forward_heads(obs)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    out[mode + "_us"] = round(us, 1)
    out[mode + "_tflops_equiv_fp32"] = round(2.0 * E * V * (5 * F1 + F1 * F2 + F2 * (4 + V)) / us / 1e6, 1)
print(json.dumps(out))
