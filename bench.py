#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the fused RIS-VEC step on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config c2|c3|c4|c5|big]

One "step" = one pass of the hot path over the whole resident env batch: every env advances by one
`Environ.step()` with its RIS cascaded channel gains recomputed from h_r and theta in the same launch
(`risvec_step_fused`, the "everything every step" mode of SURVEY 8d).  Workload at N=1: BASELINE.json
configs[2] (32 768 envs x 8 vehicles x 64 RIS elements, fp32/complex64, synthetic channel draws); N>1: the
same per GPU (weak scaling), env ids sharded by rank with no data-path collective; the north-star's
joint-observation all-gather for the global critic (RCCL) runs on a side stream every `--gather-every` steps.

With `--gpus N` > 1 and no WORLD_SIZE in the environment this process is only a launcher: before touching a GPU
it starts N rank processes (`python -m torch.distributed.run`, one per GPU, rendezvous on 127.0.0.1) and exits
with their code.  It never runs fewer ranks than asked: too few devices, a world size that differs from
`--gpus`, or a gather that cannot be constructed end the run with a non-zero exit code.

Rank 0 prints ONE JSON line with the driver's contract fields plus `roofline`, `cpu_baseline` and `legs` (the
other BASELINE configs measured in the same process: configs[1], a configs[3] shard, configs[4], and the
headline kernel on a working set far beyond the 256 MiB Infinity Cache).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

# (envs per GPU, vehicles, RIS elements, mode, default gather cadence for N > 1)
CONFIGS = {
    "c2": (4096, 8, 36, "fused", 32),       # BASELINE configs[1]
    "c3": (32768, 8, 64, "fused", 32),      # BASELINE configs[2]  (default; the metric's config)
    "c4": (8192, 8, 64, "fused", 1),        # BASELINE configs[3]: 65 536 envs over 8 GPUs, gather every step
    "c5": (32768, 16, 256, "bcd", 32),      # BASELINE configs[4]
    "big": (262144, 8, 64, "fused", 32),    # 1.36 GB per step: nothing is served by the Infinity Cache
}

# the reference itself (imported read-only) timed in the build container -- the box the
# reference can run on.  tools/time_scalar_vs_reference.py re-measures these next to the scalar-loop port.
REFERENCE_MEASURED = {
    "host": "build container: 1 core of an 8-vCPU Intel Xeon @ 2.10 GHz, Python 3.10.12, NumPy 2.2.6",
    "config": "1 env, V=8, M=64",
    "step_only_env_steps_per_s": 5.5e3,
    "gains_plus_step_env_steps_per_s": 2.35e3,
    "fused_equivalent_env_steps_per_s": 769.0,
    "fused_equivalent_definition": "renew_positions + compute_parms + update_channel_gains + step (SURVEY 6)",
    "source": "SURVEY.md section 6; tools/time_scalar_vs_reference.py (port/reference time ratio 0.98-1.01)",
}


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


def workload_name(E: int, V: int, M: int, mode: str, world: int) -> str:
    """Which BASELINE.json config (if any) the shape is."""
    if (E, V, M, mode) == (4096, 8, 36, "fused"):
        return "BASELINE configs[1]"
    if (E, V, M, mode) == (32768, 8, 64, "fused"):
        return "BASELINE configs[2]"
    if (E, V, M, mode) == (8192, 8, 64, "fused"):
        return "BASELINE configs[3]" if world == 8 else "a BASELINE configs[3] shard (8 192 envs per GPU, %d GPU%s)" % (
            world, "" if world == 1 else "s")
    if (E, V, M, mode) == (32768, 16, 256, "bcd"):
        return "BASELINE configs[4]"
    return "custom shape (no BASELINE config)"


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


# ---------------------------------------------------------------------------------------------- CPU baseline
def usable_cores() -> int:
    """Host cores this process may really use: the affinity mask, cut by a cgroup CPU quota if there is one."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def _scalar_replica(V, M, seconds, seed, q):
    from oracle import risvec_scalar as sc
    t0 = time.perf_counter()
    n = sc.time_env_steps(V, M, seconds, seed)
    q.put((n, time.perf_counter() - t0))


def cpu_baseline(V: int, M: int, budget_s: float = 18.0) -> dict:
    """"Reference NumPy step() timed on the same box's host cores" (north_star), for the work of one GPU bench
    step (update_channel_gains + step).  The reference cannot travel to the GPU box, so what runs here is its
    port: (i) `oracle/risvec_scalar.py`, the structure-faithful single-env scalar-loop restatement whose outputs
    equal the reference's bit for bit and whose run time is the reference's to +-3 % (checked in the build
    container), on ONE core and on ALL usable cores as process-per-core replicas -- the latter is `value`; (ii)
    `oracle/risvec_oracle.py`, vectorised over 2 048 envs on one thread (the strongest form of the port).
    Must run BEFORE the first GPU call of this process: the replicas are forked."""
    import multiprocessing as mp
    from oracle import risvec_oracle as orc
    from oracle import risvec_scalar as sc
    try:
        torch.set_num_threads(1)
    except Exception:
        pass
    cores = min(usable_cores(), 256)
    t_leg = budget_s / 3.0
    # (i-a) one core
    n1 = sc.time_env_steps(V, M, t_leg, 0)
    one_core = n1 / t_leg
    # (i-b) every usable core, one replica process each (fork: no GPU context exists yet)
    all_cores = one_core
    if cores > 1:
        ctx = mp.get_context("fork")
        q = ctx.Queue()
        procs = [ctx.Process(target=_scalar_replica, args=(V, M, t_leg, 100 + i, q)) for i in range(cores)]
        for pr in procs:
            pr.start()
        res = [q.get(timeout=t_leg * 4 + 120) for _ in procs]
        for pr in procs:
            pr.join(timeout=60)
        all_cores = float(sum(n / t for n, t in res))
    # (ii) vectorised restatement, one thread
    rng = np.random.default_rng(0)
    p = orc.OracleParams.yaml_effective()
    b = orc.phase_R(M)
    Es = 2048
    pos = np.stack([rng.uniform(0, 400, (Es, V)), rng.uniform(0, 400, (Es, V))], -1)
    dist, _, h_r = orc.geometry(pos, M)
    th = np.exp(1j * orc.possible_angles(3)[rng.integers(0, 8, (Es, M))])
    act = rng.uniform(0, 1, (Es, 2, V))
    partner, ng = synthetic_groups(Es, V, rng)
    st = dict(buf=np.full((Es, V), 3.0), q=np.zeros(Es))

    def run(n_iter):
        t0 = time.perf_counter()
        for _ in range(n_iter):
            gain = orc.gain_free(th, h_r, b, dist)
            arr = rng.poisson(p.rate, st["buf"].shape)
            o = orc.step(st["buf"], st["q"], gain, act, partner.astype(np.int64), ng.astype(np.int64), arr, p)
            st["buf"], st["q"] = o["data_buf"], o["mec_q"]
        return time.perf_counter() - t0
    run(1)
    t1 = run(2) / 2
    n = max(2, int(t_leg / t1))
    vec = Es * n / run(n)
    return dict(value=all_cores, unit="env-steps/s", cores=cores, kind="port",
                sample="oracle/risvec_scalar.py (structure-faithful scalar-loop port of Environment.py, float64, "
                       "bit-identical outputs, run time within 3 %% of the reference's): update_channel_gains + step of one "
                       "env at V=%d, M=%d, %d process-per-core replicas x %.0f s each" % (V, M, cores, t_leg),
                single_core_value=one_core, single_core_sample="the same on 1 core, %.0f s (%d env-steps)" % (t_leg, n1),
                vectorised_value=vec, vectorised_sample="oracle/risvec_oracle.py: gain_free + step vectorised over %d envs x "
                                                        "%d steps, float64 NumPy, 1 thread" % (Es, n),
                host_cores=os.cpu_count(), reference_measured=REFERENCE_MEASURED)


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


def yardstick_read(n_bytes: int, device) -> dict | None:
    """What the memory system of THIS device delivers to the simplest possible kernels (float4 reads with a running
    sum, tools/membench) on a buffer of the step's read size, re-read every launch like the step's h_r / theta: the best
    of the round-1 grid-strided reader and the round-2 variants (a contiguous chunk per workgroup, 8 / 16 loads in flight
    per lane, with and without the non-temporal hint) -- the ceiling a streaming kernel can be held to."""
    import ctypes as C
    path = os.path.join(ROOT, "tools", "membench", "libmembench.so")
    if not os.path.isfile(path):
        return None
    lib = C.CDLL(path)
    lib.membench_read.argtypes = [C.c_void_p, C.c_longlong, C.c_void_p, C.c_int, C.c_void_p]
    have2 = hasattr(lib, "membench_read2")
    if have2:
        lib.membench_read2.argtypes = [C.c_void_p, C.c_longlong, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    n4 = max(1, n_bytes // 16)
    src = torch.empty(n4 * 4, dtype=torch.float32, device=device).normal_()
    stream = torch.cuda.current_stream(device).cuda_stream
    sink = torch.empty(8192 * 256, dtype=torch.float32, device=device)
    cases = [("grid-strided, 4 loads in flight", lambda bl: lib.membench_read(src.data_ptr(), n4, sink.data_ptr(), bl, stream))]
    if have2:
        for u, ch, nt in ((8, 1, 0), (16, 1, 0), (16, 0, 0), (8, 1, 1)):
            cases.append(("%s, %d loads in flight%s" % ("chunk per workgroup" if ch else "grid-strided", u, ", non-temporal" if nt else ""),
                          lambda bl, u=u, ch=ch, nt=nt: lib.membench_read2(src.data_ptr(), n4, sink.data_ptr(), bl, u, ch, nt, stream)))
    best = None
    n = 100 if n_bytes < (1 << 30) else 20
    for label, fn in cases:
        for blocks in (1024, 2048, 4096):
            for _ in range(10):
                fn(blocks)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            a.record()
            for _ in range(n):
                fn(blocks)
            b.record()
            torch.cuda.synchronize()
            t = a.elapsed_time(b) * 1e-3 / n
            if best is None or t < best[0]:
                best = (t, "%s, %d workgroups" % (label, blocks))
    return dict(kernel="tools/membench: float4 read + running sum (%s)" % best[1], bytes=n4 * 16, us=best[0] * 1e6,
                GBps=n4 * 16 / best[0] / 1e9)


# ---------------------------------------------------------------------------------------------- one measured case
class Case:
    """One workload resident on one GPU: builds the env and the pre-marshalled launchers, steps it."""

    def __init__(self, E, V, M, mode, device, rank, world, start, opts, gather_every):
        from ris_vec_marl_amd import dist as rdist
        self.E, self.V, self.M, self.mode, self.device = E, V, M, mode, device
        self.opts = opts
        env = self.env = build_env(E, V, M, device, seed=0, env_offset=start)
        # BCD every step: theta lives as the sweep's candidate indices between sweeps (VecEnviron.lazy_theta: the sweep
        # does not write the complex64 tensor, the fused step expands the indices; same outputs bit for bit)
        # (not with --steer: that kernel reads the complex64 tensor, which would then be rebuilt from the indices every step)
        env.lazy_theta = mode == "bcd" and not getattr(opts, "steer", False) and not os.environ.get("RISVEC_BENCH_EAGER_THETA")
        rng = np.random.default_rng(1234 + rank)
        action = torch.from_numpy(rng.uniform(0, 1, (E, 2, V)).astype(np.float32)).to(device)
        partner_np, ng_np = synthetic_groups(E, V, rng)
        partner = torch.from_numpy(partner_np).to(device)
        n_groups = torch.from_numpy(ng_np).to(device)
        if mode == "cached":
            env.update_channel_gains()
        self.multi = 1
        self.ring_fused = False
        self.last_kernel = ""
        self.gather, self.gather_every = None, gather_every
        self.gather_note = "n/a (1 GPU)"
        if world > 1 and gather_every > 0:
            # no try/except: a run that cannot build the collective must fail, not report a number without it
            self.gather = rdist.JointObsGather(E, V, device, n_buffers=GATHER_SLOTS)
            self.gather.start(env.tensors["obs"])
            self.gather.wait()
            self.gather_note = ("joint obs [E_local,5V] fp32 all-gather (torch.distributed backend %r%s) every %d step(s), "
                                "side stream, staged, %d slots in flight, %.2f MB per rank per gather"
                                % (self.gather.backend, " = RCCL" if self.gather.backend == "nccl" else "",
                                   gather_every, GATHER_SLOTS, E * 20 * V / 1e6))
        elif world > 1:
            self.gather_note = "off (--gather-every 0)"

        fused, bcd = mode != "cached", mode == "bcd"
        full = not opts.lean
        self.full, self.fused, self.bcd = full, fused, bcd
        # arguments validated and marshalled once; each call is then a single C-ABI launch
        grouper = None
        if opts.noma:
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
        direct = False
        if opts.replay:
            from ris_vec_marl_amd import VecReplayBuffer, marshal_actions
            power_raw = torch.from_numpy(rng.uniform(-1, 1, (E, V, 2)).astype(np.float32)).to(device)
            probs = torch.from_numpy(rng.dirichlet(np.ones(V), (E, V)).astype(np.float32)).to(device)
            a_store = torch.empty(E, V * (V + 2), device=device)
            floor = float(env.cpu_share_floor)
            # Default: NO marshalling launch -- the env, the grouping and the ring read the policy outputs in place
            # (RISVEC_STEP_POLICY_ACTION, risvec_noma_group_raw, risvec_replay_store_policy).  --marshal keeps the
            # round-1 form (one more launch writing action_env / p_off01 / action_store).
            direct = not getattr(opts, "marshal", False) and not opts.policy and mode != "sarl"
            marshal = None if direct else (lambda: marshal_actions(power_raw, probs, floor, out=(action, p_off01, a_store)))   # noqa: E731
            if opts.policy:
                from ris_vec_marl_amd import BatchedPolicy
                policy = BatchedPolicy(V, 5, 512, 256, device=device, seed=rank, env_offset=start)

                def marshal():                 # policy forward + sample + marshal; outputs land in the bound tensors
                    policy.choose_action(env.tensors["obs"], grouper.mask, cpu_share_floor=floor, want_onehot=False,
                                         out=(action, p_off01, a_store))
            if marshal is not None:
                marshal()
            replay = VecReplayBuffer(16 * E, 5, V + 2, V, device=device)
            # Default (round 3): the transition store rides in the step kernel (risvec_step_ring) where that form exists;
            # RISVEC_BENCH_SEPARATE_STORE=1 keeps the two-launch form (step, then k_replay_store) for A/Bs.
            self.ring_fused = (direct and not opts.steer and V in (4, 8, 16) and mode in ("fused", "cached")
                               and not os.environ.get("RISVEC_BENCH_SEPARATE_STORE")
                               and (mode == "cached" or (V, M) in ((8, 64), (8, 36), (8, 40), (4, 16), (16, 64), (16, 256))))
            if not self.ring_fused:
                store = replay.bind_store(None, None if direct else a_store, env.tensors["metrics"], env.tensors["reward"],
                                          env.tensors["obs"], grouper.mask, policy_out=(power_raw, probs) if direct else None)
        if grouper is None:
            group = None
        elif direct:
            group = grouper.bind_group(power_raw=power_raw)
        else:
            group = grouper.bind_group(p_off01)
        if mode == "sarl":
            from ris_vec_marl_amd.sarl import SarlParams
            phase = torch.from_numpy(rng.uniform(0, 2 * np.pi, (E, M)).astype(np.float32)).to(device)
            sp = SarlParams()
            launch = env.bind_sarl_step(action, phase, None, sp, obs=full)
        elif getattr(opts, "multi", 0) > 1:
            if mode not in ("fused", "cached") or opts.noma or opts.meter or opts.steer:
                raise SystemExit("--multi T is the T-step launch of the fused gains+step path, or with --mode cached of the "
                                 "step on cached gains (no --noma/--meter/--steer)")
            T = int(opts.multi)
            actions = torch.from_numpy(rng.uniform(0, 1, (T, E, 2, V)).astype(np.float32)).to(device)
            traj = dict(reward=torch.empty(T, E, V, device=device), obs=torch.empty(T, E, V, 5, device=device),
                        metrics=torch.empty(T, E, 16, device=device)) if full else {}
            launch = env.bind_step_many(actions, partner, n_groups, None, metrics=full, obs=full, out=traj, fused=fused)
            self.multi = T
        elif self.ring_fused:
            both = env.bind_step_store(replay, power_raw, partner, n_groups, probs, grouper.mask, None, fused=fused, metrics=full,
                                       power_w=opts.meter)
            self.step_store = both
            launch = lambda: both(False, False)              # noqa: E731  (the untimed naming launch of run())
        else:
            launch = env.bind_step(power_raw if direct else action, partner, n_groups, None, fused=fused, bcd=bcd, metrics=full,
                                   power_w=opts.meter, obs=full, steer=opts.steer and fused, policy_action=direct)
        self.episode_len = 100
        meter = meter_add = None
        if opts.meter:
            if mode == "sarl" or not full:
                raise SystemExit("--meter needs the MARL step with metrics written")
            from ris_vec_marl_amd import EpisodeMeter
            meter = EpisodeMeter(env)
            meter_add = meter.bind(env)
        self.grouper, self.marshal, self.group, self.launch, self.store = grouper, marshal, group, launch, store
        self.meter, self.meter_add = meter, meter_add
        self.keep = (action, partner, n_groups, p_off01)

    def one_step(self, i: int) -> None:
        L = self.episode_len
        if self.grouper is not None:
            t = i % L
            if t == 0:
                self.grouper.begin_episode(i // L)
                self.grouper.refresh_mask()
            if self.marshal is not None:
                self.marshal()
            self.group(t)
        if self.ring_fused:                             # step + transition store in one launch
            self.step_store((i % L) == L - 1, (i % L) == 0)
        elif self.multi == 1 or i % self.multi == 0:    # --multi T: one launch advances the envs by T steps
            self.launch()
        if self.store is not None:
            self.store(done=(i % L) == L - 1, use_mask=(i % L) == 0)
        if self.meter is not None:
            if i % L == 0:
                self.meter.begin_episode()
            self.meter_add()
            if i % L == L - 1:
                self.meter.summarize()
        if self.gather is not None and i % self.gather_every == 0:
            self.gather.start(self.env.tensors["obs"])

    def run(self, steps: int, warmup: int, world: int):
        """W untimed steps, then exactly K timed steps bracketed by barrier + synchronize on both sides; the wall
        time is the MAX over ranks.  Also the HIP-event time of the K launches on the launch stream."""
        device = self.device
        from ris_vec_marl_amd import _native as N
        self.launch()                                   # untimed: names the step kernel before other launchers overwrite it
        self.last_kernel = N.last_kernel()
        for i in range(warmup):
            self.one_step(i)
        if warmup and self.group is None and self.store is None and self.meter is None:
            self.last_kernel = N.last_kernel()          # the steady-state form (e.g. theta by index from the 2nd BCD step on)
        if self.gather is not None:
            self.gather.wait()
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev0.record()
        for i in range(steps):
            self.one_step(i)
        ev1.record()
        if self.gather is not None:
            self.gather.wait()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            torch.distributed.barrier()
            on_dev = torch.distributed.get_backend() == "nccl"
            tt = torch.tensor([dt], dtype=torch.float64, device=device if on_dev else "cpu")
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            dt = float(tt.item())
        # average device time per step on the launch stream (HIP events bracket the K launches on the
        # stream they are issued on, so inter-launch gaps are included: slightly pessimistic)
        return dt, ev0.elapsed_time(ev1) / steps

    def per_env_bytes(self) -> float:
        n = algorithmic_bytes(self.V, self.M, self.mode)
        if self.opts.steer and self.fused:      # h_r (8VM) replaced by the float64 steering bases (16V)
            n += 16 * self.V - 8 * self.V * self.M
        T = getattr(self.opts, "multi", 0)
        if T > 1:
            # A T-step launch reads h_r / theta / the env's state once and writes the env's tensors once per T steps;
            # what it moves EVERY step is the action (8V) and that step's trajectory record (reward 4V + obs 20V +
            # metrics 64).  Its own algorithmic bytes per env-step, not SURVEY 8d's per-step figure (round 3: the legs
            # used to divide the per-step figure by the launch time and print a fraction > 1).
            return n / T + 32 * self.V + 64
        return n

    def kernel_name(self) -> str:
        """The kernel(s) the last launch dispatched, as the library itself reports them (risvec_last_kernel): which member
        of the fused-step family a shape / batch size takes is a dispatch decision inside librisvec.so."""
        from ris_vec_marl_amd import _native as N
        k = self.last_kernel or N.last_kernel()
        if self.mode == "bcd":
            return ("k_bcd_sweep8_pair (two lanes per env, %s) + " % ("theta by index" if self.env.lazy_theta else "theta written")) + k
        if self.mode == "sarl":
            return "k_set_phase + " + k
        return k

    def close(self):
        self.env = self.launch = self.group = self.store = self.marshal = self.grouper = self.meter = self.keep = None
        torch.cuda.empty_cache()


class _Opts:
    lean = steer = noma = replay = policy = meter = marshal = False
    multi = 0

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)


def run_leg(name, E, V, M, mode, device, rank, world, steps, warmup, gather_every=0, multi=0, yardstick=True):
    """A secondary workload measured in the same process with the same timing protocol."""
    start = rank * E
    case = (StubCase if STUB else Case)(E, V, M, mode, device, rank, world, start, _Opts(multi=multi), gather_every)
    dt, kernel_ms = case.run(steps, warmup, world)
    per_env = case.per_env_bytes()
    out = {"launch": ("one launch per step" if multi <= 1 else "T-step launch (risvec_step_fused_multi / risvec_step_multi), T = %d: "
                      "gains once per launch, queues in registers, every step's reward / obs / metrics recorded" % multi),
           "launch_amortised": multi > 1,
           "workload": "%s: %d envs/GPU x %d vehicles x %d RIS elements, %s" % (workload_name(E, V, M, mode, world), E, V, M, mode),
           "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3, "env_steps_per_s": E * world * steps / dt,
           "avg_launch_ms": kernel_ms, "algorithmic_bytes_per_env_step": per_env,
           "algorithmic_bytes_note": ("SURVEY 8d per-step figure" if multi <= 1 else
                                      "this launch's own bytes: (SURVEY 8d per-step figure) / T + 8V action + 24V + 64 record per step"),
           "roofline_frac": per_env * E / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "roofline_frac_wall": per_env * E / (dt / steps) / 1e9 / HBM_PEAK_GBS,
           "kernel": case.kernel_name(), "allgather": case.gather_note}
    if getattr(case.env, "placement", None):
        out["stream_placement"] = case.env.placement
    if mode == "bcd":
        out["bcd_candidate_evals_per_s"] = E * world * steps / dt * M * 8
        out["theta"] = ("kept as the sweep's candidate indices between sweeps (VecEnviron.lazy_theta): the sweep does not write "
                        "the complex64 tensor and the fused step expands the indices; materialised on access"
                        if case.env.lazy_theta else "complex64 tensor written by every sweep")
    if STUB:
        out["gather_ok"] = case.gather_ok
        out["env_steps_per_s"] = None
    case.close()
    # (not for T-step launches: a single-launch reader of 1.7 MB measures the launch floor the T-step form exists to avoid)
    if yardstick and rank == 0 and world == 1 and not STUB and multi <= 1:
        # the best pure float4 reader of THE SAME NUMBER OF BYTES on this device, in this process
        y = yardstick_read(int(per_env * E), device)
        if y:
            out["yardstick"] = y
            out["frac_of_yardstick"] = per_env * E / (kernel_ms * 1e-3) / 1e9 / y["GBps"]
    return out


# ---------------------------------------------------------------------------------------------- launcher (N > 1)
def launch_ranks(args, argv) -> int:
    """`python bench.py --gpus N` with no WORLD_SIZE: start N rank processes from this (GPU-clean) parent."""
    n = args.gpus
    if not args.stub:
        have = torch.cuda.device_count()          # counts devices without creating a context
        if have < n:
            sys.stderr.write("bench.py: --gpus %d but only %d HIP device(s) are visible; refusing to run fewer ranks "
                             "than asked\n" % (n, have))
            return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


STUB = False     # --stub: CPU self-test of the launcher / collective path (set in main)


class StubCase:
    """CPU stand-in for `Case` (--stub; tests/test_dist_gloo.py): the SAME main() -- rank processes, rendezvous, sharding,
    gather cadence and slots, barriers, max-over-ranks timing, the secondary legs and the JSON schema of the real run --
    with a no-op in place of the HIP step.  Its line is labelled `"stub": true` and carries no throughput."""

    def __init__(self, E, V, M, mode, device, rank, world, start, opts, gather_every):
        from ris_vec_marl_amd import dist as rdist
        self.E, self.V, self.M, self.mode, self.rank, self.world, self.start = E, V, M, mode, rank, world, start
        self.opts = opts
        self.fused, self.bcd, self.full = mode != "cached", mode == "bcd", True
        self.obs = torch.full((E, V, 5), float(rank), dtype=torch.float32)
        self.gather_every = gather_every
        self.gather = rdist.JointObsGather(E, V, "cpu", n_buffers=GATHER_SLOTS) if world > 1 and gather_every > 0 else None
        self.gather_note = ("off" if self.gather is None else
                            "joint obs all-gather (torch.distributed backend %r) every %d step(s), %d slots"
                            % (self.gather.backend, gather_every, GATHER_SLOTS))
        self.gather_ok = True
        self.env = type("E", (), {"lazy_theta": False})()

    def run(self, steps, warmup, world):
        joint, n = None, 0
        for i in range(warmup + steps):
            self.obs.add_(1.0)
            n += 1
            if self.gather is not None and i % self.gather_every == 0:
                joint, stamp = self.gather.start(self.obs), n
        if self.gather is not None:
            self.gather.wait()
        dt = torch.tensor([0.001 * (self.rank + 1)], dtype=torch.float64)
        if world > 1:
            torch.distributed.barrier()
            torch.distributed.all_reduce(dt, op=torch.distributed.ReduceOp.MAX)
        if joint is not None:                     # rank r's block holds r + (number of steps before the last gather)
            want = torch.arange(world, dtype=torch.float32).repeat_interleave(self.E) + stamp
            self.gather_ok = bool(torch.equal(joint[:, 0], want))
        return float(dt.item()), float(dt.item()) / max(1, steps) * 1e3

    def per_env_bytes(self):
        return algorithmic_bytes(self.V, self.M, self.mode)

    def kernel_name(self):
        return "stub (no GPU work)"

    def close(self):
        self.gather = None


GATHER_SLOTS = 4        # (stage, output) slots of the joint-observation gather: up to 4 gathers in flight


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS),
                    help="a BASELINE.json shape: c2 = configs[1], c3 = configs[2] (default), c4 = a configs[3] shard (8 192 envs "
                         "per GPU; with --gpus 8 it IS configs[3]), c5 = configs[4] (BCD every step), big = 262 144 envs")
    ap.add_argument("--envs-per-gpu", type=int, default=None)
    ap.add_argument("--veh", type=int, default=None)
    ap.add_argument("--ris", type=int, default=None)
    ap.add_argument("--gather-every", type=int, default=None,
                    help="N>1: all-gather the joint observation every k steps on a side stream (0 = never).  Default 32 at "
                         "32 768 envs/GPU (one gather moves 5.2 MB per rank over point-to-point xGMI links, several env "
                         "steps' worth of time, so it is amortised), 1 with --config c4 (1.3 MB per rank, BASELINE configs[3])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the secondary workloads (`legs` in the JSON line)")
    ap.add_argument("--lean", action="store_true", help="experiment: skip metrics and obs writes")
    ap.add_argument("--steer", action="store_true",
                    help="fused / bcd modes: the steering form of the gain+step kernel (RISVEC_STEP_STEER): h_r is not "
                         "read, the cascade is the polynomial sum_m theta_m b_m z^m evaluated by Horner in float64 from "
                         "the 16-byte steering base of each vehicle.  Only meaningful for the reference's own physics "
                         "(h_r from compute_parms), not for arbitrary synthetic channel draws -- NOT the headline.")
    ap.add_argument("--mode", default=None, choices=["fused", "cached", "bcd", "sarl"],
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
                         "networks over all envs) instead of pre-drawn synthetic policy outputs")
    ap.add_argument("--meter", action="store_true",
                    help="f4: add every step's metrics / rewards / powers to the per-env episode accumulators "
                         "(EpisodeMeter) and reduce the episode scalars over the envs every 100 steps")
    ap.add_argument("--marshal", action="store_true",
                    help="with --replay: keep the separate marshalling launch (action_env / p_off01 / action_store written "
                         "to HBM first) instead of letting the three consumers read the policy outputs in place")
    ap.add_argument("--multi", type=int, default=0, metavar="T",
                    help="advance the envs by T steps per launch (risvec_step_fused_multi: actions [T,E,2,V], per-step "
                         "reward / obs / metrics recorded) -- the launch shape for batches whose single step is shorter "
                         "than a kernel launch; --steps is rounded up to a multiple of T")
    ap.add_argument("--stub", action="store_true", help=argparse.SUPPRESS)     # CPU launcher self-test (tests only)
    args = ap.parse_args()
    if args.policy:
        args.replay = True
    if args.replay:
        args.noma = True
    cfg = CONFIGS[args.config or "c3"]
    args.envs_per_gpu = args.envs_per_gpu or cfg[0]
    args.veh = args.veh or cfg[1]
    args.ris = args.ris or cfg[2]
    args.mode = args.mode or cfg[3]
    if args.gather_every is None:
        args.gather_every = cfg[4]
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.multi > 1:
        args.steps = -(-args.steps // args.multi) * args.multi
        args.warmup = -(-args.warmup // args.multi) * args.multi

    # ---- N > 1 without a launcher: become one, before anything touches a GPU
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to report a world size that was not asked for"
                         % (args.gpus, world))

    # ---- CPU baseline first: its process-per-core replicas are forked, which must precede the first GPU call
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.stub:
        cpu = cpu_baseline(args.veh, args.ris)

    from ris_vec_marl_amd import dist as rdist
    global STUB
    STUB = bool(args.stub)
    if STUB:
        os.environ.setdefault("RISVEC_DIST_BACKEND", "gloo")
        rank, world, local = rdist.init_from_env(backend="gloo")
        device = torch.device("cpu")
    else:
        rank, world, local = rdist.init_from_env()
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device; there is no CPU fallback")
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
    E, V, M = args.envs_per_gpu, args.veh, args.ris
    start, _ = rdist.shard_range(E * world, rank, world)
    case = (StubCase if STUB else Case)(E, V, M, args.mode, device, rank, world, start, args, args.gather_every)
    dt, kernel_ms = case.run(args.steps, args.warmup, world)
    gather_ok = getattr(case, "gather_ok", True)
    case_ring_fused = getattr(case, "ring_fused", False)
    case_placement = getattr(getattr(case, "env", None), "placement", None)    # h_r placed by measurement (streams beyond the cache)
    fused, bcd, full = case.fused, case.bcd, case.full
    per_env = case.per_env_bytes()
    gather_note = case.gather_note
    kname = case.kernel_name()
    case.close()

    # ---- secondary workloads, same process, same protocol (every rank takes part: they contain barriers)
    legs = {}
    default_shape = (E, V, M, args.mode) == CONFIGS["c3"][:4] and not (args.noma or args.lean or args.steer or args.meter)
    if default_shape and not args.no_legs:
        if world == 1:
            legs["hbm_only"] = run_leg("hbm_only", 262144, 8, 64, "fused", device, rank, world, 200, 30)
            legs["c2"] = run_leg("c2", *CONFIGS["c2"][:4], device, rank, world, 2000, 200)
            legs["c2_multi_step"] = run_leg("c2", *CONFIGS["c2"][:4], device, rank, world, 3200, 320, multi=32)
            legs["c4_shard"] = run_leg("c4_shard", *CONFIGS["c4"][:4], device, rank, world, 2000, 200)
            legs["c5"] = run_leg("c5", *CONFIGS["c5"][:4], device, rank, world, 200, 30)
            legs["cached"] = run_leg("cached", 32768, 8, 64, "cached", device, rank, world, 2000, 200)
        else:
            n_leg, w_leg = (40, 4) if STUB else (2000, 200)
            legs["c4_gather_every_1"] = run_leg("c4", *CONFIGS["c4"][:4], device, rank, world, n_leg, w_leg, gather_every=1)
            legs["c4_gather_every_32"] = run_leg("c4", *CONFIGS["c4"][:4], device, rank, world, n_leg, w_leg, gather_every=32)
            legs["c4_no_gather"] = run_leg("c4", *CONFIGS["c4"][:4], device, rank, world, n_leg, w_leg, gather_every=0)

    if rank != 0:
        return
    bytes_per_launch = per_env * E
    achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
    achieved_wall = bytes_per_launch / (dt / args.steps) / 1e9
    traffic, traffic_src = (None, None)
    if (E, V, M, args.mode) == (32768, 8, 64, "fused") and full and not args.steer and not STUB:
        traffic, traffic_src = measured_traffic("k_step_fused")
    yard = None
    if world == 1 and fused and not args.steer and not STUB:
        yard = yardstick_read(int(per_env * E), device)
    work = {"fused": "RIS cascaded gains + step()", "cached": "step() on cached gains",
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
        "config": {"workload": "%s: %d parallel envs/GPU x %d vehicles x %d RIS elements, fp32/complex64, %s every step, "
                               "metrics+obs %s, Philox arrivals" % (workload_name(E, V, M, args.mode, world), E, V, M, work,
                                                                    "written" if full else "off"),
                   "envs_per_gpu": E, "n_veh": V, "n_ris": M, "mode": args.mode, "allgather": gather_note,
                   "noma_grouping": ("device, every step, 100-step episodes (config.yaml pairing keys)"
                                     if args.noma else "synthetic fixed groups"),
                   "replay": ("%sHBM replay ring store every step (%d B per transition)"
                              % ("marshal launch + " if (args.marshal or args.policy) else
                                 ("transition written by the step kernel (risvec_step_ring), " if case_ring_fused else
                                  "policy outputs read in place, "),
                                 4 * (10 * V + V * (V + 2) + V + 1 + V * V) + 1) if args.replay else "off"),
                   "policy": "BatchedPolicy 8x(5-512-256), every step" if args.policy else "synthetic outputs",
                   "steering_form": bool(args.steer and fused),
                   "launch": ("one launch per step" if args.multi <= 1 else
                              "T-step launch, T = %d (gains once per launch, per-step records written)" % args.multi),
                   "episode_meter": ("float64 episode sums of E x (17+V) columns every step, summary every 100 steps"
                                     if args.meter else "off"),
                   "agent_steps_per_s": E * world * args.steps / dt * V},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": kname,
                     "algorithmic_bytes_per_env_step": per_env, "bytes_per_launch": bytes_per_launch,
                     "avg_launch_ms": kernel_ms,
                     # the same fraction from the wall-clock step time the driver sees (host launch gaps included)
                     "achieved_driver": achieved_wall, "frac_driver": achieved_wall / HBM_PEAK_GBS,
                     "yardstick": yard,
                     "frac_of_yardstick": (achieved / yard["GBps"]) if yard else None,
                     "note": ("working set %.0f MB < 256 MiB Infinity Cache: h_r / theta re-reads are partly served on-die; "
                              "legs.hbm_only is the same kernel on 1.36 GB" % (bytes_per_launch / 1e6))
                             if bytes_per_launch < (256 << 20) else "working set beyond the Infinity Cache"},
        "legs": legs,
    }
    if bcd:
        out["config"]["bcd_candidate_evals_per_s"] = E * world * args.steps / dt * M * 8
    out["config"]["allgather_backend"] = torch.distributed.get_backend() if world > 1 else "none"
    out["config"]["stream_placement"] = case_placement
    out["cpu_baseline"] = cpu
    if STUB:      # a self-test line can never be mistaken for a measurement
        out.update(metric="launcher self-test (no GPU work)", value=None, stub=True, data="stub", max_over_ranks_s=dt,
                   gather_ok=bool(gather_ok and all(leg.get("gather_ok", True) for leg in legs.values())),
                   env_offset_rank0=start, backend=out["config"]["allgather_backend"])
        out["config"]["agent_steps_per_s"] = None
    print(json.dumps(out))
    if STUB and not out["gather_ok"]:
        raise SystemExit(3)


if __name__ == "__main__":
    main()
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        try:                                   # every rank gets here right after the timed region
            torch.distributed.destroy_process_group()
        except Exception:
            pass
