import os, sys, json, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from bench import build_env
for (E, V, M) in ((32768, 8, 64), (32768, 16, 256)):
    env = build_env(E, V, M, torch.device("cuda:0"), 0, 0)
    for _ in range(3): env.rebuild_colsum()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(20): env.rebuild_colsum()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    nbytes = E * (8 * V * M + 16 * M)
    print(json.dumps(dict(E=E, V=V, M=M, slab=os.environ.get("RISVEC_NO_COLSUM_SLAB") is None, us=us, GBps=nbytes / us / 1e3, frac=nbytes / us / 1e3 / 8000)))
    del env
