#!/usr/bin/env python3
"""Diagnostic: where the cycles of the indexed BCD sweep go (the s_memtime build of k_bcd_sweep8_pair in the DIAGNOSTIC
library, `make -C ris_vec_marl_amd/csrc diag`): per wavefront, cycles inside the 8-coordinate chain vs the tile
epilogue (theta / index stores).  Usage: sweep_stamps.py [E V M]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KIND = "pair"
os.environ["RISVEC_SWEEP_STAMPS"] = KIND
os.environ["RISVEC_LIB"] = os.path.join(ROOT, "ris_vec_marl_amd", "csrc", "librisvec_diag.so")
sys.path.insert(0, ROOT)
import numpy as np, torch
from bench import build_env
E, V, M = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (32768, 16, 256)))
env = build_env(E, V, M, torch.device("cuda:0"), 0, 0)
env.optimize_phase_shift()                      # generic kernel: leaves the indices
for _ in range(3):
    idx = env.optimize_phase_shift(return_idx=True)
torch.cuda.synchronize()
EPW = 32                                    # envs per wavefront (two lanes per env)
a = idx.cpu().numpy().reshape(-1)[: 4 * ((E + EPW - 1) // EPW)].reshape(-1, 4).astype(np.float64)
nb = a[0, 3]
print(json.dumps(dict(kernel=KIND, E=E, V=V, M=M, waves=int(a.shape[0]), tiles=int(nb),
                      chain_cycles_per_coordinate=float(np.median(a[:, 0]) / (nb * 8)),
                      epilogue_cycles_per_coordinate=float(np.median(a[:, 1]) / (nb * 8)),
                      total_cycles_per_coordinate=float(np.median(a[:, 2]) / (nb * 8)),
                      total_us_at_100MHz_ticks=float(np.median(a[:, 2]) / 100.0))))
