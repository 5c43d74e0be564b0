#!/usr/bin/env python3
"""Capture golden input/output vectors from the REFERENCE simulator.

Runs ONLY in the CPU build container: it imports
`/root/reference/Simulation-MARL-BCD/Environment.py` by path (read-only, no
bytecode written), wraps `numpy.random.*` / `random.choice` so every draw the
reference consumes is recorded, and writes small `.npz` fixtures (data only: inputs,
draws, expected outputs) to `tests/golden/`.  The reference never ships; these
vectors are what pins `oracle/risvec_oracle.py` (and, through it, the HIP path).

    python tools/capture_golden.py            # regenerate every fixture
"""
from __future__ import annotations

import os
import random
import sys

import numpy as np

REF_DIR = "/root/reference/Simulation-MARL-BCD"
OUT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

if not os.path.isfile(os.path.join(REF_DIR, "Environment.py")):
    sys.exit("capture_golden: reference not present at %s (this tool only runs in the build container)" % REF_DIR)

sys.dont_write_bytecode = True
sys.path.insert(0, REF_DIR)
import Environment as REF  # noqa: E402  (the reference itself)

DIR_CODE = {"u": 0, "d": 1, "l": 2, "r": 3}


def lanes():
    up = [i / 2.0 for i in [400 + 3.5 / 2, 400 + 3.5 + 3.5 / 2, 800 + 3.5 / 2, 800 + 3.5 + 3.5 / 2]]
    down = [i / 2.0 for i in [400 - 3.5 - 3.5 / 2, 400 - 3.5 / 2, 800 - 3.5 - 3.5 / 2, 800 - 3.5 / 2]]
    return dict(up=up, down=down, left=list(up), right=list(down))


def make_env(V, M, b=3):
    L = lanes()
    return REF.Environ(L["down"], L["up"], L["left"], L["right"], 400, 400, V, M, b)


class Recorder:
    """Context manager: records (name, args, value) for every draw."""
    NAMES = ("randint", "uniform", "poisson", "normal", "rand", "exponential")

    def __init__(self):
        self.log = []

    def __enter__(self):
        self._orig = {n: getattr(np.random, n) for n in self.NAMES}
        self._choice = random.choice
        for n in self.NAMES:
            setattr(np.random, n, self._wrap(n, self._orig[n]))
        random.choice = self._wrap("choice", self._choice)
        return self

    def _wrap(self, name, fn):
        def inner(*a, **k):
            v = fn(*a, **k)
            self.log.append((name, a, k, v))
            return v
        return inner

    def __exit__(self, *exc):
        for n in self.NAMES:
            setattr(np.random, n, self._orig[n])
        random.choice = self._choice
        return False

    def values(self, name):
        return [v for (n, a, k, v) in self.log if n == name]


def apply_params(env, which):
    """'default' = class defaults; 'yaml' = effective shipped values
    (marl_train_bcd.py:505-508, 563-594, 750-753 applied by hand; that script
    cannot be imported - it trains at import and needs tensorboard)."""
    if which == "yaml":
        env.w_d = 1.0; env.w_e = 1.0
        env.rate = 1.0
        env.f_local_max = 3.0e9; env.cycles_per_bit = 300.0
        env.P_max = 2.0; env.bandwidth = 5.0
        env.bandwidth_hz = env.bandwidth * 1e6
        env.noise_power = env.N0_W_per_Hz * env.bandwidth_hz
        env.power_scale = 0.7
        env.qos_enable = True; env.R_min_bpsHz = 0.15; env.D_max_s = 0.12; env.qos_penalty = 1.5


def vehicles_state(env):
    pos = np.array([v.position for v in env.vehicles], dtype=np.float64)
    direc = np.array([DIR_CODE[v.direction] for v in env.vehicles], dtype=np.int64)
    vel = np.array([v.velocity for v in env.vehicles], dtype=np.float64)
    return pos, direc, vel


# ----------------------------------------------------------------------------
def capture_reset(V, n_env, seed):
    spawn = np.zeros((n_env, V, 3), dtype=np.int64)
    buf0 = np.zeros(n_env, dtype=np.int64)
    pos = np.zeros((n_env, V, 2)); direc = np.zeros((n_env, V), dtype=np.int64)
    vel = np.zeros((n_env, V)); dbuf = np.zeros((n_env, V))
    np.random.seed(seed); random.seed(seed)
    for e in range(n_env):
        env = make_env(V, 16)
        with Recorder() as rec:
            env.make_new_game()
        ints = [v for (n, a, k, v) in rec.log if n in ("randint", "choice")]
        names = [n for (n, a, k, v) in rec.log if n in ("randint", "choice")]
        it = iter(zip(names, ints))
        for r in range(V // 4):
            _, ind = next(it)
            for s in range(4):
                _, coord = next(it); _, velv = next(it)
                spawn[e, 4 * r + s] = (ind if s == 0 else 0, coord, velv)
        for j in range(V % 4):
            _, ind = next(it)
            n, ch = next(it); assert n == "choice"
            _, coord = next(it); _, velv = next(it)
            spawn[e, 4 * (V // 4) + j] = (ind + 4 * "dulr".index(ch), coord, velv)
        _, b0 = next(it)
        buf0[e] = b0
        assert next(it, None) is None
        pos[e], direc[e], vel[e] = vehicles_state(env)
        dbuf[e] = env.DataBuf
    return dict(spawn_ints=spawn, buf0=buf0, pos=pos, direc=direc, vel=vel, data_buf=dbuf)


def renew_one_by_one(env):
    """Call the reference's renew_positions() for one vehicle at a time (vehicles
    are independent there) so each draw can be attributed to its vehicle."""
    allv = env.vehicles
    V = len(allv)
    u = np.zeros((V, 8)); n = np.zeros(V, dtype=np.int64)
    for v in range(V):
        env.vehicles = [allv[v]]
        with Recorder() as rec:
            env.renew_positions()
        d = rec.values("uniform")
        assert len(rec.log) == len(d) and len(d) <= 8
        u[v, :len(d)] = d; n[v] = len(d)
    env.vehicles = allv
    return u, n


def capture_mobility(V, n_env, T, seed):
    np.random.seed(seed); random.seed(seed)
    pos = np.zeros((n_env, T + 1, V, 2)); direc = np.zeros((n_env, T + 1, V), dtype=np.int64)
    vel = np.zeros((n_env, V)); u = np.zeros((n_env, T, V, 8)); nd = np.zeros((n_env, T, V), dtype=np.int64)
    for e in range(n_env):
        env = make_env(V, 16)
        env.make_new_game()
        pos[e, 0], direc[e, 0], vel[e] = vehicles_state(env)
        for t in range(T):
            u[e, t], nd[e, t] = renew_one_by_one(env)
            pos[e, t + 1], direc[e, t + 1], _ = vehicles_state(env)
    return dict(pos=pos, direc=direc, vel=vel, u_turn=u, n_used=nd)


def random_theta(M, b, rng):
    k = rng.integers(0, 2 ** b, size=M)
    ang = np.linspace(0, 2 * np.pi, 2 ** b, endpoint=False)[k]
    return np.cos(ang) + 1j * np.sin(ang), k


def scatter_vehicles(env, rng, n_moves):
    """Move the reference vehicles a random number of slow steps to spread them."""
    for _ in range(n_moves):
        env.renew_positions()


def capture_geometry_gain(V, M, n_env, seed):
    np.random.seed(seed); random.seed(seed)
    rng = np.random.default_rng(seed)
    out = dict(pos=[], theta=[], dist=[], ang=[], h_r=[], gain=[], b=None)
    for e in range(n_env):
        env = make_env(V, M)
        env.make_new_game()
        scatter_vehicles(env, rng, int(rng.integers(0, 250)))
        env.compute_parms()
        th, _ = random_theta(M, 3, rng)
        env.elements_phase_shift_complex[:] = th
        env.update_channel_gains()
        p, _, _ = vehicles_state(env)
        out["pos"].append(p); out["theta"].append(th)
        out["dist"].append(env.distances_R_i.copy()); out["ang"].append(env.angles_R_i.copy())
        out["h_r"].append(env.phases_R_i.copy()); out["gain"].append(env.channel_gains.copy())
        out["b"] = env.phase_R.copy()
    return {k: (np.array(v) if k != "b" else v) for k, v in out.items()}


def capture_gain3gpp(n_env, seed):
    np.random.seed(seed); random.seed(seed)
    rng = np.random.default_rng(seed)
    V = 8
    res = {}
    for mode, K_dB in (("3gpp_umi", 0.0), ("3gpp_uma", 0.0), ("3gpp_umi", 6.0), ("3gpp_uma", 3.0), ("other", 0.0)):
        tag = "%s_K%g" % (mode, K_dB)
        P, U, Z, S, G = [], [], [], [], []
        for e in range(n_env):
            env = make_env(V, 16)
            env.make_new_game()
            scatter_vehicles(env, rng, int(rng.integers(0, 250)))
            env.channel_model = mode
            env.rician_K_dB = K_dB
            with Recorder() as rec:
                env.update_channel_gains()
            p, _, _ = vehicles_state(env)
            u = np.array(rec.values("rand"))
            nrm = [(a, k, v) for (n, a, k, v) in rec.log if n == "normal"]
            if K_dB <= 1e-6:
                z = np.array([v / k["scale"] for (a, k, v) in nrm])
                small = np.array(rec.values("exponential"))
            else:
                z = np.array([nrm[3 * i][2] / nrm[3 * i][1]["scale"] for i in range(V)])
                small = np.array([nrm[3 * i + 1][2] ** 2 + nrm[3 * i + 2][2] ** 2 for i in range(V)])
                # also keep the standardised normals, to pin oracle.rice_power
                zre = np.array([(nrm[3 * i + 1][2] - nrm[3 * i + 1][1]["loc"]) / nrm[3 * i + 1][1]["scale"] for i in range(V)])
                zim = np.array([nrm[3 * i + 2][2] / nrm[3 * i + 2][1]["scale"] for i in range(V)])
                res.setdefault(tag + "_zre", []).append(zre); res.setdefault(tag + "_zim", []).append(zim)
            assert len(u) == V and len(z) == V and len(small) == V
            P.append(p); U.append(u); Z.append(z); S.append(small); G.append(env.channel_gains.copy())
        for k, v in (("pos", P), ("u_los", U), ("z_shadow", Z), ("small", S), ("gain", G)):
            res[tag + "_" + k] = np.array(v)
    return {k: np.array(v) for k, v in res.items()}


def capture_bcd(V, M, n_env, seed, b=3):
    np.random.seed(seed); random.seed(seed)
    rng = np.random.default_rng(seed)
    out = dict(pos=[], theta0=[], theta1=[], obj0=[], obj1=[], gain1=[], h_r=[], dist=[])
    for e in range(n_env):
        env = make_env(V, M, b)
        env.make_new_game()
        scatter_vehicles(env, rng, int(rng.integers(0, 250)))
        env.compute_parms()
        if e == 0:
            th0 = np.zeros(M, dtype=complex)          # the reference's own start state (ENV:171)
        else:
            th0, _ = random_theta(M, b, rng)
        env.elements_phase_shift_complex[:] = th0
        out["obj0"].append(env.optimize_compute_objective_function())
        env.optimize_phase_shift()
        out["obj1"].append(env.optimize_compute_objective_function())
        env.update_channel_gains()
        p, _, _ = vehicles_state(env)
        out["pos"].append(p); out["theta0"].append(th0)
        out["theta1"].append(np.array(env.elements_phase_shift_complex, dtype=complex))
        out["gain1"].append(env.channel_gains.copy())
        out["h_r"].append(env.phases_R_i.copy()); out["dist"].append(env.distances_R_i.copy())
    out = {k: np.array(v) for k, v in out.items()}
    out["b"] = env.phase_R.copy()
    out["control_bit"] = np.int64(b)
    return out


def random_groups(V, rng):
    """random partition into pairs / singles / unscheduled, random order in pairs,
    occasionally an (ignored) 3-element group."""
    perm = list(rng.permutation(V))
    groups = []
    while perm:
        r = rng.random()
        if r < 0.45 and len(perm) >= 2:
            groups.append([int(perm.pop()), int(perm.pop())])
        elif r < 0.90:
            groups.append([int(perm.pop())])
        elif r < 0.93 and len(perm) >= 3:
            groups.append([int(perm.pop()), int(perm.pop()), int(perm.pop())])
        else:
            perm.pop()            # unscheduled
    return groups


def encode_groups(groups, V):
    partner = np.full(V, -2, dtype=np.int64)
    for g in groups:
        if len(g) == 1:
            partner[g[0]] = -1
        elif len(g) == 2:
            partner[g[0]] = g[1]
            partner[g[1]] = g[0] + (1 << 16)
    return partner, len(groups)


STEP_OUT = ("reward", "global_reward", "data_buf", "data_t", "data_p", "over_power", "over_data",
            "vehicle_rate", "mec_q", "last_power_W")
METRICS = ("last_off_kbit_sum", "last_local_kbit_sum", "last_mec_queue_cycles",
           "last_backlog_kbit_mean", "last_delay_local_mean", "last_delay_edge_q_mean",
           "last_delay_edge_c_mean", "last_t_tx_mean", "last_mec_utilization",
           "last_local_util_mean", "last_qos_violation", "last_delay_mean", "last_energy_mean")


def run_step(env, action, groups):
    with Recorder() as rec:
        r = env.step(action, groups)
    arr = np.array(rec.values("poisson"), dtype=np.int64)
    assert len(rec.log) == len(arr) == env.n_veh
    o = dict(reward=np.array(r[0]), global_reward=float(r[1]), data_buf=np.array(r[2]),
             data_t=np.array(r[3]), data_p=np.array(r[4]), over_power=np.array(r[5]),
             over_data=np.array(r[6]), vehicle_rate=env.vehicle_rate.copy(),
             mec_q=float(env.mec_queue_cycles), last_power_W=np.array(env.last_power_W))
    o["metrics"] = np.array([o["global_reward"]] + [float(getattr(env, n)) for n in METRICS])
    return o, arr


def capture_step(V, n, seed, which):
    np.random.seed(seed); random.seed(seed)
    rng = np.random.default_rng(seed)
    env = make_env(V, 16)
    env.make_new_game()
    apply_params(env, which)
    rec = {k: [] for k in ("data_buf0", "mec_q0", "gain", "action", "partner", "n_groups", "arrivals", "metrics") + STEP_OUT}
    for i in range(n):
        mode = i % 8
        B0 = rng.uniform(0.0, 12.0, V)
        if mode == 1:
            B0 = rng.uniform(0.0, 0.5, V)              # nearly empty backlog: offload limited by backlog
        if mode == 2:
            B0 = rng.uniform(50.0, 400.0, V)           # heavy backlog: D_max violations
        if mode == 3:
            B0[rng.integers(0, V)] = 0.0
        Q0 = 0.0 if mode in (0, 1) else float(rng.uniform(0, 6e6))
        gain = 10 ** rng.uniform(-13.5, -9.5, V)
        if mode == 4:
            gain[:] = 0.0                               # fresh env: all-zero gains
        if mode == 5:
            j, k = rng.choice(V, 2, replace=False); gain[j] = gain[k]   # exact tie in a pair
        action = rng.uniform(-0.25, 1.3, (2, V))
        if mode == 6:
            action = rng.uniform(0.0, 1.0, (2, V))
        if mode == 7:
            action[0, rng.integers(0, V)] = 0.0
        groups = random_groups(V, rng)
        if mode == 5:
            groups = [[int(j), int(k)]] + [[int(q)] for q in range(V) if q not in (j, k)]
        env.DataBuf = B0.copy(); env.mec_queue_cycles = Q0; env.channel_gains = gain.copy()
        o, arr = run_step(env, action.copy(), groups)
        partner, ng = encode_groups(groups, V)
        for k, v in (("data_buf0", B0), ("mec_q0", Q0), ("gain", gain), ("action", action),
                     ("partner", partner), ("n_groups", ng), ("arrivals", arr), ("metrics", o["metrics"])):
            rec[k].append(v)
        for k in STEP_OUT:
            rec[k].append(o[k])
    return {k: np.array(v) for k, v in rec.items()}


def capture_trajectory(V, M, n_ep, n_step, refresh_every, bcd_every, seed, which="yaml"):
    """a13-a15: the driver's call protocol (marl_train_bcd.py:545, 1268-1271,
    1307-1313, 1601-1611) with a random policy standing in for the agents."""
    np.random.seed(seed); random.seed(seed)
    rng = np.random.default_rng(seed)
    env = make_env(V, M)
    with Recorder() as rec0:
        env.make_new_game()
    apply_params(env, which)
    out = dict(pos0=None, theta_seq=[], pos_seq=[], u_turn=[], policy=[], partner=[], n_groups=[],
               arrivals=[], obs=[], reward=[], global_reward=[], gain=[], mec_q=[], refresh_mask=[], bcd_mask=[])
    p0, d0, v0 = vehicles_state(env)
    out["pos0"], out["direc0"], out["vel0"], out["data_buf0"] = p0, d0, v0, env.DataBuf.copy()
    # make theta non-zero first: with the all-zero start every candidate ties for m=0
    th, _ = random_theta(M, 3, rng)
    env.elements_phase_shift_complex[:] = th
    out["theta0"] = th
    for ep in range(n_ep):
        refreshed = ep % refresh_every == 0
        if refreshed:
            u, _ = renew_one_by_one(env)
            env.compute_parms()
        else:
            u = np.zeros((V, 8))
        out["u_turn"].append(u); out["refresh_mask"].append(refreshed)
        out["pos_seq"].append(vehicles_state(env)[0])
        for st in range(n_step):
            do_bcd = st % bcd_every == 0
            if do_bcd:
                env.optimize_phase_shift()
                env.update_channel_gains()
            out["bcd_mask"].append(do_bcd)
            out["theta_seq"].append(np.array(env.elements_phase_shift_complex, dtype=complex))
            out["gain"].append(env.get_channel_gains().copy())
            obs = np.array([[env.DataBuf[i] / 10, env.data_t[i] / 10, env.data_p[i] / 10,
                             env.over_data[i] / 10, env.vehicle_rate[i] / 20] for i in range(V)])
            out["obs"].append(obs)
            pol = rng.uniform(-1.2, 1.2, (V, 2))
            act = np.zeros((2, V))
            fl = max(0.0, min(float(env.cpu_share_floor), 0.95))
            for i in range(V):
                c = np.clip(pol[i], -0.999, 0.999)
                act[0, i] = (c[0] + 1) / 2
                act[1, i] = max((c[1] + 1) / 2, fl)
            groups = random_groups(V, rng)
            o, arr = run_step(env, act, groups)
            partner, ng = encode_groups(groups, V)
            out["policy"].append(pol); out["partner"].append(partner); out["n_groups"].append(ng)
            out["arrivals"].append(arr); out["reward"].append(o["reward"])
            out["global_reward"].append(o["global_reward"]); out["mec_q"].append(o["mec_q"])
    res = {k: np.array(v) for k, v in out.items()}
    res["b"] = env.phase_R.copy()
    res["shape"] = np.array([V, M, n_ep, n_step, refresh_every, bcd_every])
    return res


def capture_sarl(V, M, n, seed):
    """f1: Simulation-SARL/Environment.py step(action_power, action_phase) (SENV:321-359)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("sarl_reference_env", "/root/reference/Simulation-SARL/Environment.py")
    SREF = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(SREF)            # NB the module seeds np.random(1234) at import (SENV:7)
    np.random.seed(seed); random.seed(seed)
    rng = np.random.default_rng(seed)
    L = lanes()
    env = SREF.Environ(L["down"], L["up"], L["left"], L["right"], 400, 400, V, M, 3)
    env.make_new_game()
    rec = {k: [] for k in ("pos", "data_buf0", "action_power", "action_phase", "arrivals", "theta", "h_r", "dist",
                           "reward_mean", "data_buf", "data_t", "data_p", "over_power", "over_data", "vehicle_rate")}
    for i in range(n):
        if i % 16 == 0:
            for _ in range(int(rng.integers(1, 40))):
                env.renew_positions()
            env.compute_parms()
        mode = i % 4
        B0 = rng.uniform(0.0, 12.0, V)
        if mode == 1:
            B0 = rng.uniform(0.0, 0.6, V)                 # drained buffers: over_data / over_power branch
        if mode == 2:
            B0 = rng.uniform(2.0, 3.5, V)
        power = rng.uniform(0.0, 1.0, (2, V))
        if mode == 3:
            power[1] = rng.uniform(0.0, 0.05, V)          # weak local CPU
        phase = rng.uniform(0.0, 2 * np.pi, M)
        env.DataBuf = B0.copy()
        with Recorder() as r:
            out = env.step(power.copy(), phase.copy())
        arr = np.array(r.values("poisson"), dtype=np.int64)
        assert len(arr) == V
        rec["pos"].append(vehicles_state(env)[0]); rec["data_buf0"].append(B0); rec["action_power"].append(power)
        rec["action_phase"].append(phase); rec["arrivals"].append(arr)
        rec["theta"].append(np.array(env.elements_phase_shift_complex, dtype=complex))
        rec["h_r"].append(env.phases_R_i.copy()); rec["dist"].append(env.distances_R_i.copy())
        rec["reward_mean"].append(float(out[0])); rec["data_buf"].append(np.array(out[1]))
        rec["data_t"].append(np.array(out[2])); rec["data_p"].append(np.array(out[3]))
        rec["over_power"].append(np.array(out[4])); rec["over_data"].append(np.array(out[5]))
        rec["vehicle_rate"].append(env.vehicle_rate.copy())
    res = {k: np.array(v) for k, v in rec.items()}
    res["b"] = env.phase_R.copy()
    return res


def save(name, d):
    path = os.path.join(OUT_DIR, name)
    np.savez_compressed(path, **d)
    print("%-32s %8.1f KB" % (name, os.path.getsize(path) / 1024))


def main():
    os.makedirs(OUT_DIR, exist_ok=True)
    for V in (4, 6, 8, 16):
        save("reset_%d.npz" % V, capture_reset(V, 64, 100 + V))
    for V in (4, 8):
        save("mobility_%d.npz" % V, capture_mobility(V, 6, 700, 200 + V))
    for (V, M, n) in ((4, 16, 8), (8, 36, 8), (8, 64, 8), (16, 256, 2)):
        save("geometry_gain_%d_%d.npz" % (V, M), capture_geometry_gain(V, M, n, 300 + V + M))
    save("gain3gpp.npz", capture_gain3gpp(12, 400))
    for (V, M, n) in ((4, 16, 6), (8, 36, 6), (8, 64, 4), (16, 256, 2)):
        save("bcd_%d_%d.npz" % (V, M), capture_bcd(V, M, n, 500 + V + M))
    save("bcd_4_16_b2.npz", capture_bcd(4, 16, 4, 777, b=2))
    for V in (4, 8, 16):
        for which in ("default", "yaml"):
            save("step_%d_%s.npz" % (V, which), capture_step(V, 384, 600 + V, which))
    save("trajectory_8_36.npz", capture_trajectory(8, 36, 7, 40, 5, 100, 900))
    save("sarl_step_8_40.npz", capture_sarl(8, 40, 192, 1100))
    save("sarl_step_4_16.npz", capture_sarl(4, 16, 128, 1101))


if __name__ == "__main__":
    main()
