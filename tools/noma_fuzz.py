#!/usr/bin/env python3
"""Extended randomised check of the pairing stage beyond 8 vehicles (tests/test_noma_hip.py::fuzz_beyond_8_vehicles over
more seeds and envs than the test suite runs).  Usage: noma_fuzz.py [n_seeds [E]]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_noma_hip import fuzz_beyond_8_vehicles

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 24
E = int(sys.argv[2]) if len(sys.argv) > 2 else 48
t0 = time.time()
checked = dense = 0
for seed in range(n_seeds):
    c, d = fuzz_beyond_8_vehicles(seed, E)
    checked, dense = checked + c, dense + d
    print(json.dumps(dict(seed=seed, ok=True, elapsed_s=round(time.time() - t0, 1))), flush=True)
print(json.dumps(dict(noma_fuzz="ok", env_steps_checked=checked, solves_with_15_or_more_matchable_users=dense)))
