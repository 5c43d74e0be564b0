#!/usr/bin/env python3
"""Build-container check of the CPU baseline: `oracle/risvec_scalar.py` (the structure-faithful
scalar-loop restatement bench.py times on the GPU box) against the REFERENCE itself, imported
read-only from /root/reference -- same outputs (to float64 rounding) and the same run time (the
survey's bar: within +-20 %).  Prints one JSON line; the figures are quoted in DESIGN.md and carried
by bench.py as `cpu_baseline.reference_measured`.

    python tools/time_scalar_vs_reference.py [--seconds 4]
"""
from __future__ import annotations

import argparse
import json
import os
import platform
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = "/root/reference/Simulation-MARL-BCD"
if not os.path.isfile(os.path.join(REF_DIR, "Environment.py")):
    sys.exit("reference not present at %s (this tool only runs in the build container)" % REF_DIR)
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, REF_DIR)
import Environment as REF  # noqa: E402  (the reference itself)

from oracle import risvec_oracle as orc  # noqa: E402
from oracle import risvec_scalar as sc  # noqa: E402


def build_pair(V, M, seed=0):
    L = orc.default_lanes()
    np.random.seed(seed)
    import random
    random.seed(seed)
    ref = REF.Environ(L["down"], L["up"], L["left"], L["right"], 400, 400, V, M, 3)
    ref.make_new_game()
    ref.renew_positions()
    ref.compute_parms()
    ref.Random_phase()
    p = orc.OracleParams.yaml_effective()
    ref.bandwidth = p.bandwidth; ref.noise_power = p.noise_power; ref.P_max = p.P_max
    ref.f_local_max = p.f_local_max; ref.cycles_per_bit = p.cycles_per_bit; ref.rate = p.rate
    ref.w_d, ref.w_e = p.w_d, p.w_e
    ref.R_min_bpsHz, ref.D_max_s, ref.qos_penalty = p.R_min_bpsHz, p.D_max_s, p.qos_penalty
    mine = sc.ScalarEnv(V, M, p, ref.elements_phase_shift_complex.copy(), ref.phases_R_i.copy(), ref.phase_R.copy(),
                        np.array(ref.distances_R_i, dtype=np.float64), ref.DataBuf.copy(), ref.mec_queue_cycles)
    return ref, mine, p


def run(V, M, seconds):
    ref, mine, p = build_pair(V, M)
    rng = np.random.default_rng(1)
    perm = rng.permutation(V)
    groups = [[int(perm[2 * k]), int(perm[2 * k + 1])] for k in range(V // 4)] + [[int(u)] for u in perm[2 * (V // 4):]]
    # (1) same outputs: drive both with the same actions; the reference draws its own arrivals from
    # numpy.random, which are read back from its data_r and injected into the restatement
    worst = 0.0
    for _ in range(50):
        a = rng.uniform(-0.1, 1.2, (2, V))
        ref.update_channel_gains(); mine.update_channel_gains()
        worst = max(worst, float(np.max(np.abs(mine.gain - ref.channel_gains) / ref.channel_gains)))
        r_ref = ref.step(a.copy(), groups)
        r_me = mine.step(a.copy(), groups, np.array(ref.data_r, dtype=np.int64))
        for x, y in zip(r_ref, r_me):
            worst = max(worst, float(np.max(np.abs(np.asarray(x) - np.asarray(y)) / (np.abs(np.asarray(x)) + 1e-9))))
        m_ref = [ref.last_off_kbit_sum, ref.last_local_kbit_sum, ref.last_mec_queue_cycles, ref.last_backlog_kbit_mean,
                 ref.last_delay_local_mean, ref.last_delay_edge_q_mean, ref.last_delay_edge_c_mean, ref.last_t_tx_mean,
                 ref.last_mec_utilization, ref.last_local_util_mean, ref.last_qos_violation, ref.last_delay_mean,
                 ref.last_energy_mean]
        m_me = mine.metrics14(float(r_me[1]))[1:]
        worst = max(worst, float(np.max(np.abs(np.array(m_ref) - m_me) / (np.abs(np.array(m_ref)) + 1e-9))))

    # (2) same run time, interleaved slices so clock drift hits both alike
    def t_ref(n):
        t0 = time.perf_counter()
        for _ in range(n):
            ref.update_channel_gains()
            ref.step(rng.uniform(0, 1, (2, V)), groups)
        return time.perf_counter() - t0

    def t_me(n):
        t0 = time.perf_counter()
        for _ in range(n):
            mine.update_channel_gains()
            mine.step(rng.uniform(0, 1, (2, V)), groups, rng.poisson(p.rate, V))
        return time.perf_counter() - t0

    def t_ref_step(n):
        t0 = time.perf_counter()
        for _ in range(n):
            ref.step(rng.uniform(0, 1, (2, V)), groups)
        return time.perf_counter() - t0

    def t_me_step(n):
        t0 = time.perf_counter()
        for _ in range(n):
            mine.step(rng.uniform(0, 1, (2, V)), groups, rng.poisson(p.rate, V))
        return time.perf_counter() - t0

    acc = dict(ref=0.0, me=0.0, ref_step=0.0, me_step=0.0)
    n_tot = 0
    t_end = time.perf_counter() + seconds
    t_ref(20); t_me(20)
    while time.perf_counter() < t_end:
        acc["ref"] += t_ref(50); acc["me"] += t_me(50)
        acc["ref_step"] += t_ref_step(50); acc["me_step"] += t_me_step(50)
        n_tot += 50
    return dict(V=V, M=M, max_rel_diff=worst,
                reference_gain_plus_step_per_s=n_tot / acc["ref"], scalar_port_gain_plus_step_per_s=n_tot / acc["me"],
                reference_step_only_per_s=n_tot / acc["ref_step"], scalar_port_step_only_per_s=n_tot / acc["me_step"],
                time_ratio_port_over_reference=acc["me"] / acc["ref"],
                time_ratio_step_only=acc["me_step"] / acc["ref_step"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=4.0)
    args = ap.parse_args()
    out = dict(host="%s, %d logical CPUs, 1 thread used" % (platform.processor() or platform.machine(), os.cpu_count()),
               python=platform.python_version(), numpy=np.__version__,
               shapes=[run(8, 64, args.seconds), run(8, 36, args.seconds), run(16, 256, args.seconds / 2)])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
