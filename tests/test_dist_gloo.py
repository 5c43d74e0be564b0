"""world_size-2 `gloo` test of the multi-GPU layer on CPU: env ids shard by rank with no
data-path collective, the ONE exchange is the joint-observation all-gather, and results
are independent of the sharding (RNG keyed by global env id).  The per-shard env work is
done by the CPU oracle here (the product has no CPU path); what is under test is
`ris_vec_marl_amd.dist` and the sharding contract."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as td
import torch.multiprocessing as mp

from oracle import risvec_oracle as orc

E, V, SEED, STEPS = 64, 8, 2024, 3


def shard_obs(lo, hi):
    """obs [hi-lo, V, 5] after STEPS oracle steps for global env ids lo..hi-1."""
    ids = np.arange(lo, hi)
    rng = np.random.default_rng(7)                    # same full-batch inputs on every rank
    gain = 10 ** rng.uniform(-13, -10, (E, V))
    action = rng.uniform(0, 1, (E, 2, V))
    p = orc.OracleParams.yaml_effective()
    buf, q = np.full((hi - lo, V), 3.0), np.zeros(hi - lo)
    partner = np.full((hi - lo, V), -1); ng = np.full(hi - lo, V)
    o = None
    for s in range(STEPS):
        arr = orc.philox_arrivals(ids, V, s, SEED, p.rate)
        o = orc.step(buf, q, gain[lo:hi], action[lo:hi], partner, ng, arr, p)
        buf, q = o["data_buf"], o["mec_q"]
    return o["obs"].astype(np.float32)


def episode_summary(lo, hi):
    """[3, 21] mean / min / max over envs lo..hi-1 of a fixed synthetic per-env table."""
    per_env = torch.from_numpy(np.random.default_rng(11).normal(size=(E + 1, 21)))[lo:hi]
    return torch.stack([per_env.mean(0), per_env.min(0).values, per_env.max(0).values])


def worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from ris_vec_marl_amd import dist as rdist
    r, w, _ = rdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    lo, n = rdist.shard_range(E + 1, rank, world)
    assert n == (33, 32)[rank]
    summ_lo, summ_n = lo, n
    lo, n = rdist.shard_range(E, rank, world)
    obs_local = torch.from_numpy(shard_obs(lo, lo + n))
    g = rdist.JointObsGather(n, V, "cpu")
    a = g.start(obs_local); g.wait()
    b = g.start(obs_local * 2); g.wait()             # second buffer of the double buffer
    joint = rdist.gather_joint_obs(obs_local)
    summ = rdist.combine_episode_summary(episode_summary(summ_lo, summ_lo + summ_n), summ_n)   # unequal shards: 33 + 32 envs
    if rank == 0:
        out.put((a.numpy().copy(), b.numpy().copy(), joint.numpy().copy(), summ.numpy().copy()))
    td.barrier()
    td.destroy_process_group()


def test_joint_obs_allgather_world2():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    a, b, joint, summ = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    whole = shard_obs(0, E).reshape(E, 5 * V)         # one rank owning every env
    assert a.shape == (E, 5 * V)
    np.testing.assert_array_equal(a, whole)           # sharded == unsharded, bit for bit
    np.testing.assert_array_equal(b, whole * 2)
    np.testing.assert_array_equal(joint, whole)
    want = episode_summary(0, E + 1).numpy()             # the episode summary of one rank owning every env
    np.testing.assert_allclose(summ[0], want[0], rtol=1e-13, atol=1e-15)
    np.testing.assert_array_equal(summ[1:], want[1:])


def test_single_process_gather_is_a_copy():
    from ris_vec_marl_amd import dist as rdist
    x = torch.arange(2 * 3 * 5, dtype=torch.float32).reshape(2, 3, 5)
    y = rdist.gather_joint_obs(x)
    assert y.shape == (2, 15) and torch.equal(y, x.reshape(2, 15))
    s3 = torch.arange(63, dtype=torch.float64).reshape(3, 21)
    assert torch.equal(rdist.combine_episode_summary(s3, 5), s3)


# ---------------------------------------------------------------------------- bench.py's own launcher
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv, env=None, timeout=300):
    import subprocess
    import sys
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), cwd=ROOT, env=e,
                          capture_output=True, text=True, timeout=timeout)


def test_bench_launcher_spawns_the_ranks_it_was_asked_for():
    """`python bench.py --gpus 2` with no WORLD_SIZE must start 2 rank processes itself (here: gloo on CPU with
    the HIP step stubbed out -- what is under test is the launcher, the rendezvous, the sharding, the gather
    cadence and the max-over-ranks timing), and rank 0 prints the one JSON line."""
    import json
    r = _bench("--gpus", "2", "--stub", "--steps", "5", "--warmup", "2", "--envs-per-gpu", "16", "--gather-every", "2")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["stub"] is True and d["value"] is None          # a self-test line can never be mistaken for a measurement
    assert d["n_gpus"] == 2 and d["backend"] == "gloo" and d["gather_ok"] is True
    assert abs(d["max_over_ranks_s"] - 0.002) < 1e-12          # MAX over ranks (rank 1 reported 2 ms)


def test_bench_scale_command_with_eight_ranks():
    """The command the driver's SCALE step issues -- `python bench.py --gpus 8` with the default workload -- end to end on
    CPU (gloo, the HIP step stubbed out): 8 rank processes, 32 768 envs per rank, the three BASELINE configs[3] legs
    (8 192 envs per rank with the all-gather every step / every 32 steps / off), every gather's contents checked on
    rank 0, and the JSON schema of the real line (`n_gpus`, `legs`, the backend named)."""
    import json
    r = _bench("--gpus", "8", "--stub", "--steps", "6", "--warmup", "2", timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["stub"] is True and d["value"] is None and d["n_gpus"] == 8 and d["scaling"] == "weak"
    assert d["config"]["envs_per_gpu"] == 32768 and d["config"]["allgather_backend"] == "gloo"
    assert set(d["legs"]) == {"c4_gather_every_1", "c4_gather_every_32", "c4_no_gather"}
    for name, leg in d["legs"].items():
        assert "BASELINE configs[3]" in leg["workload"] and leg["gather_ok"] is True, name
        assert leg["env_steps_per_s"] is None
    assert "every 1 step" in d["legs"]["c4_gather_every_1"]["allgather"] and d["legs"]["c4_no_gather"]["allgather"] == "off"
    assert d["gather_ok"] is True and abs(d["max_over_ranks_s"] - 0.008) < 1e-12      # MAX over 8 ranks
    for key in ("metric", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data", "roofline"):
        assert key in d, key


def test_bench_never_silently_runs_fewer_ranks():
    """No device in this container: `--gpus 2` must exit non-zero (not fall back to one rank), and a WORLD_SIZE that
    differs from --gpus is refused as well."""
    r = _bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    assert r.returncode != 0 and "refusing" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    r = _bench("--gpus", "4", "--stub", "--steps", "1", "--warmup", "0",
               env=dict(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"))
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_bench_workload_labels():
    import importlib.util
    spec = importlib.util.spec_from_file_location("risvec_bench", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b.workload_name(4096, 8, 36, "fused", 1) == "BASELINE configs[1]"
    assert b.workload_name(32768, 8, 64, "fused", 1) == "BASELINE configs[2]"
    assert b.workload_name(8192, 8, 64, "fused", 8) == "BASELINE configs[3]"
    assert "shard" in b.workload_name(8192, 8, 64, "fused", 1)
    assert b.workload_name(32768, 16, 256, "bcd", 1) == "BASELINE configs[4]"
    assert "custom" in b.workload_name(1000, 8, 64, "fused", 1)
    assert b.algorithmic_bytes(8, 64) == 5188 and b.algorithmic_bytes(8, 36) == 3172 and b.algorithmic_bytes(16, 256, "bcd") == 37956
    assert b.usable_cores() >= 1
