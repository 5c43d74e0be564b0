"""Structure-faithful scalar-loop restatement of ONE reference environment -- TEST / BASELINE
INFRASTRUCTURE, never imported by the product (`ris_vec_marl_amd/`).

`risvec_oracle.py` restates the reference vectorised over a leading env axis (the fast form of the
port).  This module restates the same two methods the GPU benchmark step covers the way the
reference itself executes them -- one env, Python loops over vehicles / RIS elements / groups,
`math`/`cmath` scalars, small NumPy vectors and one `np.mean` per logged scalar -- so that its
run time is the reference's run time (interpreter-bound) and can stand in for
"reference NumPy step() timed on the same box's host cores" on a GPU box the reference cannot
travel to.  Pinned: outputs against the vectorised oracle and the golden vectors
(`tests/test_oracle_golden.py::test_scalar_loop_*`), timing against the imported reference in the
build container (`tools/time_scalar_vs_reference.py`, result quoted in DESIGN.md).

Reference: Simulation-MARL-BCD/Environment.py (ENV) update_channel_gains "free" ENV:263-273,
compute_data_rate ENV:331-372, step ENV:547-731.
"""
from __future__ import annotations

import math
from typing import List, Sequence

import numpy as np

from .risvec_oracle import ALPHA1, ALPHA2, RO, OracleParams, distance_B_R, effective_floor


class ScalarEnv:
    """State of one env as the reference keeps it: per-vehicle float64 vectors + Python scalars."""

    def __init__(self, n_veh: int, M: int, p: OracleParams, theta: np.ndarray, h_r: np.ndarray, b: np.ndarray,
                 dist: np.ndarray, data_buf: np.ndarray, mec_q: float = 0.0):
        self.V, self.M, self.p = int(n_veh), int(M), p
        self.theta = np.asarray(theta, dtype=np.complex128)          # elements_phase_shift_complex [M]
        self.h_r = np.asarray(h_r, dtype=np.complex128)              # phases_R_i [V, M]
        self.b = np.asarray(b, dtype=np.complex128)                  # phase_R [M]
        self.dist = np.asarray(dist, dtype=np.float64)               # distances_R_i [V]
        self.d_br = distance_B_R()
        self.gain = np.zeros(self.V)
        self.buf = np.array(data_buf, dtype=np.float64)
        self.q = float(mec_q)
        self.rate = np.zeros(self.V)
        self.data_t = np.zeros(self.V)
        self.data_p = np.zeros(self.V)
        self.over_data = np.zeros(self.V)
        self.log = {}                                                # the 13 last_* scalars

    # ENV:263-273 -- V x M Python loop of complex multiply-adds, then the path-loss division
    def update_channel_gains(self) -> None:
        for v in range(self.V):
            acc = 0
            for m in range(self.M):            # attribute + double index per term, as the reference pays it
                term = self.theta[m] * self.h_r[v][m] * self.b[m]
                acc += term
            casc = (RO * acc) / (math.sqrt(self.dist[v] ** ALPHA1) * math.sqrt(self.d_br ** ALPHA2))
            self.gain[v] = np.abs(casc) ** 2

    # ENV:331-372 -- loop over the groups; users in no 1-/2-element group keep rate 0
    def compute_data_rate(self, power: np.ndarray, groups: Sequence[Sequence[int]]) -> np.ndarray:
        g = self.gain
        out = np.zeros(self.V)
        n0 = self.p.noise_power
        for grp in groups:
            frac = 1.0 / max(1, len(groups))
            if len(grp) == 1:
                u = grp[0]
                out[u] = frac * math.log2(1 + power[0, u] * g[u] / n0)
            elif len(grp) == 2:
                a, c = grp[0], grp[1]
                near, far = (a, c) if g[a] > g[c] else (c, a)
                out[far] = frac * math.log2(1 + power[0, far] * g[far] / (power[0, near] * g[far] + n0))
                out[near] = frac * math.log2(1 + power[0, near] * g[near] / n0)
        return out

    # ENV:547-731
    def step(self, action: np.ndarray, groups: Sequence[Sequence[int]], arrivals: Sequence[int]):
        p, V = self.p, self.V
        proj = np.clip(action, 0.0, None) * p.power_scale                       # ENV:555-561
        for v in range(V):
            s = proj[:, v].sum()
            if s > 1.0:
                proj[:, v] /= (s + 1e-12)
        power = proj * p.P_max
        self.rate[:] = self.compute_data_rate(power, groups)                    # ENV:565-570
        self.data_t[:] = self.rate * p.time_fast * p.bandwidth * 1000.0
        share_cpu = np.maximum(np.clip(action[1, :], 0.0, 1.0), effective_floor(p.cpu_share_floor))   # ENV:572-580
        f_loc = share_cpu * p.f_local_max
        cpb = float(p.cycles_per_bit)
        before = self.buf.copy()                                                # ENV:585-592
        before_cyc = before * 1000.0 * cpb
        cap = f_loc * p.time_fast
        used = np.minimum(cap, before_cyc)
        done = used / (cpb * 1000.0)
        self.data_p[:] = done
        left = np.maximum(0.0, before - self.data_p)                            # ENV:595-601
        off = np.minimum(self.data_t, left)
        t_tx = np.divide(off, self.rate * p.bandwidth * 1000.0 + 1e-12)
        e_in = off * 1000.0 * cpb                                               # ENV:604-610
        q_before = self.q
        self.q += e_in.sum()
        served = min(p.f_edge_max * p.time_fast, self.q)
        self.q -= served
        lg = self.log
        lg["off_kbit_sum"] = float(off.sum())                                   # ENV:612-614
        lg["local_kbit_sum"] = float(done.sum())
        lg["mec_queue_cycles"] = float(self.q)
        self.buf -= (self.data_p + off)                                         # ENV:617-619
        self.buf = np.maximum(0.0, self.buf)
        lg["backlog_kbit_mean"] = float(self.buf.mean())
        eps = 1e-12                                                             # ENV:622-633
        d_loc = np.maximum(0.0, before_cyc - e_in) / (f_loc + eps)
        frac = e_in / (e_in.sum() + eps)
        d_q = frac * (q_before / (p.f_edge_max + eps))
        d_c = e_in / (p.f_edge_max + eps)
        delay = d_loc + t_tx + d_q + d_c
        for _ in range(2):                                                      # the reference assigns these twice (ENV:636-646)
            lg["delay_local_mean"] = float(np.mean(d_loc))
            lg["delay_edge_q_mean"] = float(np.mean(d_q))
            lg["delay_edge_c_mean"] = float(np.mean(d_c))
            lg["t_tx_mean"] = float(np.mean(t_tx))
        lg["backlog_kbit_mean"] = float(np.mean(before))                        # ENV:649 overwrites :619
        lg["mec_utilization"] = float(served / (p.f_edge_max * p.time_fast + 1e-12))      # ENV:652-656
        lg["local_util_mean"] = float(np.mean(used / (cap + 1e-12)))
        e_tx = power[0, :] * t_tx                                               # ENV:659-666
        e_loc = p.k * (f_loc ** 2) * used
        energy = e_tx + e_loc
        self.power_eq = np.vstack([e_tx / p.time_fast, e_loc / p.time_fast])
        pen = np.zeros(V, dtype=float)                                          # ENV:669-677
        if p.qos_enable:
            viol = (np.asarray(self.rate, dtype=float) < float(p.R_min_bpsHz)) | (np.asarray(delay, dtype=float) > float(p.D_max_s))
            if np.any(viol):
                pen = float(p.qos_penalty) * viol.astype(float)
            lg["qos_violation"] = float(np.mean(viol.astype(float)))
        _d = float(np.mean(delay))                                              # ENV:691-692 (computed, unused)
        _e = float(np.mean(energy))
        reward = np.clip(-(float(p.w_d) * delay + float(p.w_e) * energy) - pen, -float(p.reward_clip), float(p.reward_clip))
        for _ in range(2):                                                      # ENV:706-711 (assigned twice)
            lg["delay_mean"] = float(np.mean(delay))
            lg["energy_mean"] = float(np.mean(energy))
        for v in range(V):                                                      # ENV:717-719
            self.buf[v] += arrivals[v] * p.time_fast * 1000
        g_reward = np.mean(reward)                                              # ENV:721-729
        over_power = np.maximum(0.0, (power[0, :] + power[1, :]) - p.P_max)
        return reward, g_reward, self.buf, self.data_t, self.data_p, over_power, self.over_data

    def metrics14(self, g_reward: float) -> np.ndarray:
        lg = self.log
        return np.array([g_reward, lg["off_kbit_sum"], lg["local_kbit_sum"], lg["mec_queue_cycles"], lg["backlog_kbit_mean"],
                         lg["delay_local_mean"], lg["delay_edge_q_mean"], lg["delay_edge_c_mean"], lg["t_tx_mean"],
                         lg["mec_utilization"], lg["local_util_mean"], lg.get("qos_violation", 0.0), lg["delay_mean"],
                         lg["energy_mean"]])


def groups_from_partner(partner: np.ndarray, n_groups: int) -> List[List[int]]:
    """The batched encoding back to the reference's list of lists (pairs in listed order)."""
    groups: List[List[int]] = []
    for v, q in enumerate(partner):
        q = int(q)
        if q == -1:
            groups.append([v])
        elif 0 <= q < (1 << 16):
            groups.append([v, q])
    while len(groups) < int(n_groups):
        groups.append([])                  # groups of other sizes still count in G (ENV:341)
    return groups


def time_env_steps(V: int, M: int, seconds: float, seed: int = 0, with_gains: bool = True) -> int:
    """Step one scalar-loop env for ~`seconds` of wall time; returns the number of env-steps done.
    One step = update_channel_gains (the "free" cascade) + step(), i.e. the work of one GPU bench step."""
    import time
    from .risvec_oracle import geometry, phase_R, possible_angles
    rng = np.random.default_rng(seed)
    p = OracleParams.yaml_effective()
    pos = np.stack([rng.uniform(0, 400, (1, V)), rng.uniform(0, 400, (1, V))], -1)
    dist, _, h_r = geometry(pos, M)
    theta = np.exp(1j * possible_angles(3)[rng.integers(0, 8, M)])
    env = ScalarEnv(V, M, p, theta, h_r[0], phase_R(M), dist[0], np.full(V, 3.0))
    perm = rng.permutation(V)
    groups = [[int(perm[2 * k]), int(perm[2 * k + 1])] for k in range(V // 4)] + [[int(u)] for u in perm[2 * (V // 4):]]
    n, t_end = 0, time.perf_counter() + seconds
    if not with_gains:
        env.update_channel_gains()
    while time.perf_counter() < t_end:
        for _ in range(8):
            if with_gains:
                env.update_channel_gains()
            env.step(rng.uniform(0, 1, (2, V)), groups, rng.poisson(p.rate, V))
        n += 8
    return n
