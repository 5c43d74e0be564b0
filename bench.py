#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the fused RIS-VEC step on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one launch of the fused north-star kernel (RIS cascaded gains + step(),
`risvec_step_fused`) over the whole resident env batch: every env advances by one
`Environ.step()` with its channel gains recomputed from h_r and theta ("everything
every step" mode of SURVEY 8d).  Workload at N=1: BASELINE.json configs[2]
(32 768 envs x 8 vehicles x 64 RIS elements, fp32/complex64, synthetic channel draws);
N>1: the same per GPU (weak scaling), env ids sharded by rank with no data-path
collective; the north-star's joint-observation all-gather for the global critic runs on
a side stream every `--gather-every` steps.

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` and
`cpu_baseline`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def algorithmic_bytes(V: int, M: int, mode: str = "fused") -> int:
    """bytes one env-step must move (fp32 / complex64).
    fused : reads h_r 8VM + theta 8M + action 8V + DataBuf 4V + path-loss 4V + partner 4V +
            n_groups 4 + Q 4; writes gain 4V + DataBuf 4V + reward 4V + rate/data_t/data_p 12V +
            obs 20V + Q 4 + metrics 56                               = 8VM + 8M + 64V + 68
    cached: the same without h_r/theta and with the gain read instead of written = 60V + 68
    bcd   : fused + theta written back (h_r counted once: the sweep and the gain pass are
            separate kernels, so real traffic is ~2x h_r)            = 8VM + 16M + 64V + 68"""
    if mode == "cached":
        return 60 * V + 68
    if mode == "bcd":
        return 8 * V * M + 16 * M + 64 * V + 68
    if mode == "sarl":      # + the agent's phase row read (4M) and theta written then read (8M + 8M); 48V + 4 of step I/O
        return 8 * V * M + 20 * M + 48 * V + 4
    return 8 * V * M + 8 * M + 64 * V + 68


def synthetic_groups(E: int, V: int, rng) -> tuple:
    """V//4 random pairs + singles per env (SURVEY 8d 'synthetic inputs')."""
    partner = np.full((E, V), -1, dtype=np.int32)
    perm = np.argsort(rng.random((E, V)), axis=1).astype(np.int32)
    rows = np.arange(E)
    for k in range(V // 4):
        a, b = perm[:, 2 * k], perm[:, 2 * k + 1]
        partner[rows, a] = b
        partner[rows, b] = a + (1 << 16)
    n_groups = np.full(E, V - V // 4, dtype=np.int32)
    return partner, n_groups


def build_env(E: int, V: int, M: int, device, seed: int, env_offset: int):
    from ris_vec_marl_amd import VecEnviron, reference_lanes, apply_yaml_config
    L = reference_lanes()
    env = VecEnviron(L["down_lanes"], L["up_lanes"], L["left_lanes"], L["right_lanes"], 400, 400, V, M, 3,
                     n_envs=E, device=device, seed=seed, env_offset=env_offset)
    # YAML-effective physics: shipped config.yaml + driver Config defaults
    apply_yaml_config(env, dict(mec=dict(f_local_max=3.0e9, cycles_per_bit=300), phy=dict(bandwidth_MHz=5, P_max=2.0),
                                env=dict(rate=1), reward=dict(sample=False, w_d_fixed=1.0, w_e_fixed=1.0),
                                qos_enable=True, qos_penalty=1.5))
    env.make_new_game()                 # reference reset distribution (Philox)
    for _ in range(3):
        env.renew_positions()
    env.compute_parms()                 # h_r = reference steering vectors from geometry
    env.Random_phase()                  # theta uniform over the 2^b discrete phases
    return env


def cpu_baseline(V: int, M: int, budget_s: float = 12.0) -> dict:
    """The CPU oracle (float64 NumPy restatement of the reference) timed on this box's host
    cores for the same per-step work (cascaded gains + step): baseline only."""
    from oracle import risvec_oracle as orc
    try:
        torch.set_num_threads(1)
    except Exception:
        pass
    rng = np.random.default_rng(0)
    p = orc.OracleParams.yaml_effective()
    b = orc.phase_R(M)

    def make(Es):
        pos = np.stack([rng.uniform(0, 400, (Es, V)), rng.uniform(0, 400, (Es, V))], -1)
        dist, _, h_r = orc.geometry(pos, M)
        th = np.exp(1j * orc.possible_angles(3)[rng.integers(0, 8, (Es, M))])
        act = rng.uniform(0, 1, (Es, 2, V))
        partner, ng = synthetic_groups(Es, V, rng)
        return dict(dist=dist, h_r=h_r, th=th, act=act, partner=partner.astype(np.int64), ng=ng.astype(np.int64),
                    buf=np.full((Es, V), 3.0), q=np.zeros(Es))

    def run(st, n_iter):
        t0 = time.perf_counter()
        for _ in range(n_iter):
            gain = orc.gain_free(st["th"], st["h_r"], b, st["dist"])
            arr = rng.poisson(p.rate, st["buf"].shape)
            o = orc.step(st["buf"], st["q"], gain, st["act"], st["partner"], st["ng"], arr, p)
            st["buf"], st["q"] = o["data_buf"], o["mec_q"]
        return time.perf_counter() - t0

    # (a) vectorised over a 2048-env sample (the strongest form of the port)
    Es = 2048
    st = make(Es)
    run(st, 1)
    t1 = run(st, 2) / 2
    n = max(2, int(0.6 * budget_s / t1))
    tv = run(st, n) / n
    # (b) one env at a time, as the reference itself is driven
    s1 = make(1)
    run(s1, 5)
    t1e = run(s1, 20) / 20
    n1 = max(20, int(0.3 * budget_s / t1e))
    ts = run(s1, n1) / n1
    return dict(value=Es / tv, unit="env-steps/s", cores=1, kind="port",
                sample="oracle/risvec_oracle.py (float64 NumPy, 1 thread): gain_free+step vectorised over %d envs "
                       "x %d steps at V=%d, M=%d; single-env stepping (how the reference is driven) = %.0f env-steps/s"
                       % (Es, n, V, M, 1.0 / ts),
                single_env_value=1.0 / ts, host_cores=os.cpu_count())


def measured_traffic(kernel_prefix: str):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/*_pmc_summary.json, written by tools/summarize_pmc.py from separate --pmc
    FETCH_SIZE / WRITE_SIZE runs of this same command): FETCH_SIZE x 2 (gfx950 counts a wide
    coalesced stream at half its bytes, MI355X_MICROARCH.md) + WRITE_SIZE, KiB -> bytes."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    try:
        names = sorted(f for f in os.listdir(pdir) if f.endswith("_pmc_summary.json"))
    except OSError:
        return None, None
    for name in names:                                   # the latest round wins
        try:
            d = json.load(open(os.path.join(pdir, name)))
        except Exception:
            continue
        for k, v in d.items():
            if kernel_prefix in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
                best = ((2.0 * v["FETCH_SIZE"]["mean_kib"] + v["WRITE_SIZE"]["mean_kib"]) * 1024.0, name)
    return best if best else (None, None)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs-per-gpu", type=int, default=32768)
    ap.add_argument("--veh", type=int, default=8)
    ap.add_argument("--ris", type=int, default=64)
    ap.add_argument("--gather-every", type=int, default=32,
                    help="N>1: all-gather the joint observation every k steps on a side stream (0 = never). "
                         "At 32 768 envs/GPU one gather moves 5.2 MB per rank over point-to-point xGMI links, "
                         "several env steps' worth of time (an 8-GPU ring moves 7 x 5.2 MB into every GPU, ~0.2 ms), so "
                         "it is amortised rather than issued every step; at 32 its duty cycle stays near 20 %%.")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lean", action="store_true", help="experiment: skip metrics and obs writes")
    ap.add_argument("--steer", action="store_true",
                    help="fused / bcd modes: the steering form of the gain+step kernel (RISVEC_STEP_STEER): h_r is not "
                         "read, the cascade is the polynomial sum_m theta_m b_m z^m evaluated by Horner in float64 from "
                         "the 16-byte steering base of each vehicle.  Only meaningful for the reference's own physics "
                         "(h_r from compute_parms), not for arbitrary synthetic channel draws -- NOT the headline.")
    ap.add_argument("--mode", default="fused", choices=["fused", "cached", "bcd", "sarl"],
                    help="fused: gains+step each step (headline); cached: step only (reference cadence); "
                         "bcd: BCD sweep + gains + step each step (BASELINE config 5); "
                         "sarl: the single-agent env variant's step (SURVEY 8 f1): phases from the agent + gains + step")
    ap.add_argument("--noma", action="store_true",
                    help="also run the NOMA grouping stage (SURVEY 8 f2) before every step, with the reference's "
                         "episode structure: 100-step episodes, mask rebuilt at step 0, groups frozen in between")
    ap.add_argument("--replay", action="store_true",
                    help="full device-resident rollout step (SURVEY 8 f3): marshal synthetic policy outputs, group "
                         "(implies --noma), step, append the E transitions to the HBM replay ring")
    ap.add_argument("--policy", action="store_true",
                    help="with --replay: the actions come from the batched SAC policy (BatchedPolicy: 8 x (5-512-256) "
                         "networks over all envs, rocBLAS GEMMs + the fused sampling/marshalling kernel) instead of "
                         "pre-drawn synthetic policy outputs")
    ap.add_argument("--meter", action="store_true",
                    help="f4: add every step's metrics / rewards / powers to the per-env episode accumulators "
                         "(EpisodeMeter) and reduce the episode scalars over the envs every 100 steps")
    args = ap.parse_args()
    if args.policy:
        args.replay = True
    if args.replay:
        args.noma = True

    from ris_vec_marl_amd import dist as rdist
    rank, world, local = rdist.init_from_env()
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; there is no CPU fallback")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    E, V, M = args.envs_per_gpu, args.veh, args.ris
    start, _ = rdist.shard_range(E * world, rank, world)
    env = build_env(E, V, M, device, seed=0, env_offset=start)
    rng = np.random.default_rng(1234 + rank)
    action = torch.from_numpy(rng.uniform(0, 1, (E, 2, V)).astype(np.float32)).to(device)
    partner_np, ng_np = synthetic_groups(E, V, rng)
    partner = torch.from_numpy(partner_np).to(device)
    n_groups = torch.from_numpy(ng_np).to(device)
    if args.mode == "cached":
        env.update_channel_gains()

    gather = None
    gather_note = "n/a (1 GPU)"
    if world > 1 and args.gather_every > 0:
        try:
            gather = rdist.JointObsGather(E, V, device)
            gather.start(env.tensors["obs"]); gather.wait()
            gather_note = ("joint obs [E_local,5V] fp32 all-gather (RCCL over xGMI) every %d step(s), side stream, "
                           "staged + double-buffered" % args.gather_every)
        except Exception as ex:           # keep the env path measurable even if RCCL is unavailable
            gather = None
            gather_note = "disabled: %r" % (ex,)
    elif world > 1:
        gather_note = "off (--gather-every 0)"

    fused, bcd = args.mode != "cached", args.mode == "bcd"
    full = not args.lean

    # arguments validated and marshalled once; each call is then a single C-ABI launch
    grouper = None
    if args.noma:
        from ris_vec_marl_amd import NomaGrouper
        grouper = NomaGrouper(env)
        grouper.config.apply_yaml({"min_pair_target": 3, "mwm_backoff_rounds": 3, "qos_enable": True,
                                   "reward": {"mask_topk_start": 7, "mask_topk_end": 7, "mask_tau_q_start": 0.10,
                                              "mask_tau_q_end": 0.25, "pairing_threshold_quantile": 0.25}})
        if not fused:
            env.update_channel_gains()
        else:
            env.bind_step(action, partner, n_groups, None, fused=True, metrics=full, power_w=False, obs=full)()
        grouper.begin_episode(0); grouper.refresh_mask()
        partner, n_groups = grouper.group(action[:, 0, :].contiguous(), 0)     # state views: stable pointers
    p_off01 = action[:, 0, :].contiguous()
    marshal = store = None
    if args.replay:
        from ris_vec_marl_amd import VecReplayBuffer, marshal_actions
        power_raw = torch.from_numpy(rng.uniform(-1, 1, (E, V, 2)).astype(np.float32)).to(device)
        probs = torch.from_numpy(rng.dirichlet(np.ones(V), (E, V)).astype(np.float32)).to(device)
        a_store = torch.empty(E, V * (V + 2), device=device)
        floor = float(env.cpu_share_floor)
        marshal = lambda: marshal_actions(power_raw, probs, floor, out=(action, p_off01, a_store))   # noqa: E731
        if args.policy:
            from ris_vec_marl_amd import BatchedPolicy
            policy = BatchedPolicy(V, 5, 512, 256, device=device, seed=rank, env_offset=start)

            def marshal():                 # policy forward + sample + marshal; outputs land in the bound tensors
                policy.choose_action(env.tensors["obs"], grouper.mask, cpu_share_floor=floor, want_onehot=False,
                                     out=(action, p_off01, a_store))
        marshal()
        replay = VecReplayBuffer(16 * E, 5, V + 2, V, device=device)
        store = replay.bind_store(None, a_store, env.tensors["metrics"], env.tensors["reward"], env.tensors["obs"],
                                  grouper.mask)
    group = grouper.bind_group(p_off01) if grouper is not None else None
    if args.mode == "sarl":
        from ris_vec_marl_amd.sarl import SarlParams
        phase = torch.from_numpy(rng.uniform(0, 2 * np.pi, (E, M)).astype(np.float32)).to(device)
        sp = SarlParams()
        launch = lambda: env.sarl_step(action, phase, None, sp, obs=full)       # noqa: E731
    else:
        launch = env.bind_step(action, partner, n_groups, None, fused=fused, bcd=bcd, metrics=full,
                               power_w=args.meter, obs=full, steer=args.steer and fused)
    episode_len = 100
    meter = meter_add = None
    if args.meter:
        if args.mode == "sarl" or not full:
            raise SystemExit("--meter needs the MARL step with metrics written")
        from ris_vec_marl_amd import EpisodeMeter
        meter = EpisodeMeter(env)
        meter_add = meter.bind(env)

    def one_step(i: int) -> None:
        if grouper is not None:
            t = i % episode_len
            if t == 0:
                grouper.begin_episode(i // episode_len)
                grouper.refresh_mask()
            if marshal is not None:
                marshal()
            group(t)
        launch()
        if store is not None:
            store(done=(i % episode_len) == episode_len - 1, use_mask=(i % episode_len) == 0)
        if meter is not None:
            if i % episode_len == 0:
                meter.begin_episode()
            meter_add()
            if i % episode_len == episode_len - 1:
                meter.summarize()
        if gather is not None and i % args.gather_every == 0:
            gather.start(env.tensors["obs"])

    for i in range(args.warmup):
        one_step(i)
    if gather is not None:
        gather.wait()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        one_step(i)
    ev1.record()
    if gather is not None:
        gather.wait()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        torch.distributed.barrier()
        on_dev = torch.distributed.get_backend() == "nccl"
        tt = torch.tensor([dt], dtype=torch.float64, device=device if on_dev else "cpu")
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    # average device time per launch on the launch stream (HIP events bracket the K launches
    # on the stream they are issued on, so inter-launch gaps are included: slightly pessimistic)
    kernel_ms = ev0.elapsed_time(ev1) / args.steps

    if rank != 0:
        return
    per_env = algorithmic_bytes(V, M, args.mode)
    if args.steer and fused:            # h_r (8VM) replaced by the float64 steering bases (16V)
        per_env += 16 * V - 8 * V * M
    bytes_per_launch = per_env * E
    achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
    kname = {"fused": "k_step_fused", "cached": "k_step<", "bcd": "k_bcd", "sarl": "SarlCore"}[args.mode]
    traffic, traffic_src = (None, None)
    if (E, V, M, args.mode) == (32768, 8, 64, "fused") and full:
        traffic, traffic_src = measured_traffic(kname)
    workload = {"fused": "RIS cascaded gains + step()", "cached": "step() on cached gains",
                "bcd": "BCD sweep + gains + step()", "sarl": "SARL get_next_phase + gains + step()"}[args.mode]
    out = {
        "metric": "env-steps/sec (all agents) at 8 veh x 64 RIS",
        "value": E * world * args.steps / dt,
        "unit": "env-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[%d]: %d parallel envs/GPU x %d vehicles x %d RIS elements, "
                               "fp32/complex64, %s every step, metrics+obs %s, Philox arrivals"
                               % (4 if bcd else 2, E, V, M, workload, "written" if full else "off"),
                   "envs_per_gpu": E, "n_veh": V, "n_ris": M, "mode": args.mode, "allgather": gather_note,
                   "noma_grouping": ("device, every step, 100-step episodes (config.yaml pairing keys)"
                                     if args.noma else "synthetic fixed groups"),
                   "replay": ("marshal + HBM replay ring store every step (%d B per transition)"
                              % (4 * (10 * V + V * (V + 2) + V + 1 + V * V) + 1) if args.replay else "off"),
                   "policy": "BatchedPolicy 8x(5-512-256), every step" if args.policy else "synthetic outputs",
                   "steering_form": bool(args.steer and fused),
                   "episode_meter": ("float64 episode sums of E x (17+V) columns every step, summary every 100 steps"
                                     if args.meter else "off"),
                   "agent_steps_per_s": E * world * args.steps / dt * V},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": "k_step_steer" if (args.steer and fused) else {"fused": "k_step_fused_pipe<8,64,2,MarlCore>" if (V, M) == (8, 64) else "k_step_fused*",
                                "cached": "k_step", "bcd": "k_bcd_lane + k_step_fused*",
                                "sarl": "k_set_phase + k_step_fused_pipe<..,SarlCore>"}[args.mode],
                     "algorithmic_bytes_per_env_step": per_env, "bytes_per_launch": bytes_per_launch,
                     "avg_launch_ms": kernel_ms},
    }
    if bcd:
        out["config"]["bcd_candidate_evals_per_s"] = E * world * args.steps / dt * M * 8
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(V, M)
    elif world > 1:
        out["cpu_baseline"] = None
    print(json.dumps(out))


if __name__ == "__main__":
    main()
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        try:                                   # every rank gets here right after the timed region
            torch.distributed.destroy_process_group()
        except Exception:
            pass
