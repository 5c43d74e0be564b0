"""GPU tests of the entry points the BASELINE configs are actually benchmarked through
(round 2): `step(bcd=True)` / `bind_step(bcd=True)` -> `risvec_step_fused_bcd` (config 5), the
on-device joint-observation gather (config 4) and the objects built with the DEFAULT device.
Everything goes through the C ABI; the oracle is the checker."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from oracle import risvec_oracle as orc  # noqa: E402  (checker)
from tests.test_hip_parity import (RT, c128, check_step, cpu, make_vec, put_complex, random_step_inputs, record, snap,  # noqa: E402
                                   step_mask)


# ---------------------------------------------------------------------------- config 5 entry point
def _bcd_step_reference(t, th_in, B0, Q0, action, partner, ng, arrivals, p, bbit):
    """oracle for ONE step(bcd=True): bcd_sweep -> gain_free -> step on the device's own float32 tensors
    (Environment.py:208-231, 255-273, 547-731 in the order marl_train_bcd.py:1307-1312, 1611 calls them)."""
    h, b = c128(t["h_r"]), c128(t["b"])
    dist = cpu(t["dist_r"]).astype(np.float64)
    o_th, o_idx = orc.bcd_sweep(th_in, h, b, dist, bbit)
    gap = orc.bcd_margin(th_in, h, b, bbit)
    safe = np.minimum.accumulate(gap, axis=1) > 1e-9          # a flipped decision taints the rest of its sweep
    return o_th, o_idx, safe, h, b


@pytest.mark.parametrize("V,M,E", [(16, 256, 96), (8, 64, 300)])
@pytest.mark.parametrize("bound", [False, True])
def test_step_bcd_entry_vs_oracle(V, M, E, bound):
    """The launch path BASELINE configs[4] is benchmarked through (bench.py --mode bcd): consecutive
    `step(bcd=True)` / `bind_step(bcd=True)` calls against oracle `bcd_sweep -> gain_free -> step`, including the
    bookkeeping the number depends on: the column sums stay cached, sweeps 2.. start from the sum the previous
    sweep left in s_sum (`_ssum_sweeps` counts them), and Random_phase() / a direct theta write announced with
    invalidate_colsum() force the re-summing pass."""
    bbit = 3
    rng = np.random.default_rng(100 * V + M + int(bound))
    p = orc.OracleParams.yaml_effective()
    env = make_vec(E, V, M, b=bbit, seed=17, yaml=True)
    env.make_new_game()
    env.renew_positions()
    env.compute_parms()
    env.Random_phase()
    t = env.tensors
    B0 = rng.uniform(0, 12, (E, V)).astype(np.float32)
    t["data_buf"].copy_(torch.from_numpy(B0))
    action, partner, ng, _ = random_step_inputs(E, V, rng)
    a_dev = torch.from_numpy(action.astype(np.float32)).cuda()
    pt, ngt = torch.from_numpy(partner.astype(np.int32)).cuda(), torch.from_numpy(ng.astype(np.int32)).cuda()
    arr_dev = torch.zeros(E, V, dtype=torch.int32, device="cuda:0")
    launch = env.bind_step(a_dev, pt, ngt, arr_dev, fused=True, bcd=True) if bound else None
    assert env._colsum_valid and env._ssum_sweeps == 0

    worst = dict(theta=0.0, gain=0.0, reward=0.0)

    def one_step(expect_sweeps):
        th_in = snap(c128(t["theta"]), bbit)
        Bq, Qq = cpu(t["data_buf"]).astype(np.float64), cpu(t["mec_q"]).astype(np.float64)
        arrivals = rng.poisson(1.0, (E, V))
        arr_dev.copy_(torch.from_numpy(arrivals.astype(np.int32)))
        a_dev.mul_(0.97)                                             # inputs are re-read on every launch
        act = cpu(a_dev).astype(np.float64)
        o_th, o_idx, safe, h, b = _bcd_step_reference(t, th_in, Bq, Qq, act, partner, ng, arrivals, p, bbit)
        if bound:
            launch()
            out = (t["reward"], t["metrics"][:, 0], t["data_buf"])
        else:
            out = env.step(a_dev, pt, ngt, arr_dev, fused=True, bcd=True)
        assert env._colsum_valid and env._ssum_sweeps == expect_sweeps
        # (1) theta: same decisions as the oracle wherever float64 noise cannot flip one
        th1 = c128(t["theta"])
        record("BCD decisions left out of the comparison (fraction; float64 margin below 1e-9)", 1.0 - safe.mean())
        assert safe.mean() > 0.999
        err_th = np.abs(th1 - o_th)[safe]
        worst["theta"] = max(worst["theta"], float(err_th.max()))
        assert err_th.max() <= 1.5e-7
        # the sum the sweep left behind is the sum of what it stored (the next sweep starts from it)
        S = cpu(t["s_sum"]); S = S[:, 0] + 1j * S[:, 1]
        want_S = np.sum(snap(th1, bbit) * (h.sum(axis=1) * b[None, :]), axis=1)
        np.testing.assert_allclose(S, want_S, rtol=1e-10, atol=1e-10)
        # (2) gains from the NEW theta (same launch sequence): 1e-5 + the float32 cancellation floor of an M-term sum
        pl = cpu(t["pl"]).astype(np.float64)
        img = np.einsum("em,evm,m->ev", th1, h, b)
        gain = pl * np.abs(img) ** 2
        g_dev = cpu(t["gain"]).astype(np.float64)
        floor = pl * 2 * np.abs(img) * (3 * 6e-8 * M)
        assert (np.abs(g_dev - gain) <= RT * gain + floor).all()
        worst["gain"] = max(worst["gain"], float(np.max((np.abs(g_dev - gain) - floor) / gain)))
        # and against the oracle's own sweep result for envs whose whole sweep is safe
        env_safe = safe.all(axis=1)
        assert env_safe.mean() > 0.999                                # (observed: every env)
        g_orc = orc.gain_free(o_th, h, b, cpu(t["dist_r"]).astype(np.float64))
        img_o = np.einsum("em,evm,m->ev", o_th, h, b)
        # dist_r is float32 on the device: pl carries 2.2 * 6e-8 of that
        assert (np.abs(g_dev - g_orc) <= RT * g_orc + pl * 2 * np.abs(img_o) * (3 * 6e-8 * M))[env_safe].all()
        # (3) step outputs with the device's gains as input
        o = orc.step(Bq, Qq, g_dev, act, partner, ng, arrivals, p)
        near_qos, near_other = step_mask(o, partner, g_dev, Qq)
        okr = check_step(env, out, o, Bq, p, near_qos, near_other)
        record("samples left out of the reward comparison (fraction; discontinuities of step())", 1.0 - okr.mean())
        assert okr.mean() > 0.999
        rel = np.abs(cpu(out[0]) - o["reward"])[okr] / np.maximum(np.abs(o["reward"][okr]), 1e-3)
        worst["reward"] = max(worst["reward"], float(rel.max()))

    for k in range(4):                                   # sweep 1 re-sums, sweeps 2..4 take the cached-sum path
        one_step(k + 1)
    env.Random_phase()
    assert env._ssum_sweeps == 0 and env._colsum_valid   # theta changed, geometry did not
    one_step(1)
    one_step(2)
    put_complex(t["theta"], np.exp(1j * rng.uniform(0, 2 * np.pi, (E, M))))    # arbitrary (non-candidate) phases
    env.invalidate_colsum()
    assert env._ssum_sweeps == 0 and not env._colsum_valid
    one_step(1)                                          # rebuilds c_col and re-sums
    one_step(2)
    print("step(bcd=True) V=%d M=%d bound=%s: max |dtheta| %.2e, gain rel err above floor %.2e, reward rel err %.2e"
          % (V, M, bound, worst["theta"], worst["gain"], worst["reward"]))


def test_step_bcd_flag_translation_equals_separate_calls():
    """risvec_step_fused_bcd with REUSE flags == optimize_phase_shift() + step(fused=True), bit for bit, on every
    combination of (cached column sums, cached sum) the host bookkeeping can produce."""
    E, V, M = 257, 8, 64
    rng = np.random.default_rng(7)
    action, partner, ng, arrivals = random_step_inputs(E, V, rng)
    a = torch.from_numpy(action.astype(np.float32)).cuda()
    pt, ngt = torch.from_numpy(partner.astype(np.int32)).cuda(), torch.from_numpy(ng.astype(np.int32)).cuda()
    ar = torch.from_numpy(arrivals.astype(np.int32)).cuda()
    envs = []
    for fused_entry in (True, False):
        env = make_vec(E, V, M, seed=5, yaml=True)
        env.make_new_game(); env.renew_positions(); env.compute_parms(); env.Random_phase()
        seen = []
        for k in range(5):
            if k == 3:
                env.invalidate_colsum()                  # cold: column sums rebuilt + re-summed
            if fused_entry:
                env.step(a, pt, ngt, ar, fused=True, bcd=True)
            else:
                env.optimize_phase_shift()
                env.step(a, pt, ngt, ar, fused=True)
            seen.append((env._colsum_valid, env._ssum_sweeps))
        envs.append((env, seen))
    assert envs[0][1] == envs[1][1] == [(True, 1), (True, 2), (True, 3), (True, 1), (True, 2)]
    for k in ("theta", "gain", "reward", "data_buf", "mec_q", "metrics", "obs", "s_sum"):
        assert torch.equal(envs[0][0].tensors[k], envs[1][0].tensors[k]), k


# ---------------------------------------------------------------------------- config 4: the gather, on the device
def _check_gather(g, env, a, pt, ngt):
    """start/wait twice (both halves of the double buffer) against obs.reshape while the env keeps stepping on
    the main stream: the result must be the observation AT start(), not a later one."""
    t = env.tensors
    E, V = env.n_envs, env.n_veh
    for rnd in range(4):
        env.step(a, pt, ngt, None, fused=True)
        want = t["obs"].reshape(E, 5 * V).clone()
        out = g.start(t["obs"])
        for _ in range(6):                               # overwrite obs several times while the gather is in flight
            env.step(a, pt, ngt, None, fused=True)
        g.wait()
        assert out.shape == (g.world * E, 5 * V)
        assert torch.equal(out[:E], want), rnd
        assert not torch.equal(t["obs"].reshape(E, 5 * V), want)      # the env really moved on meanwhile
    assert g.n_started == 4 and g.i == 0                 # two buffers, used alternately


def _check_gathers_in_flight(g, env, a, pt, ngt, k=3):
    """k gathers started back to back, each after another env step, BEFORE the first wait: every returned buffer must
    hold the observation at ITS start(); then one more round through the same slots (slot reuse orders itself)."""
    t = env.tensors
    E, V = env.n_envs, env.n_veh
    for rnd in range(2):
        wants, outs = [], []
        for _ in range(k):
            env.step(a, pt, ngt, None, fused=True)
            wants.append(t["obs"].reshape(E, 5 * V).clone())
            outs.append(g.start(t["obs"]))
            env.step(a, pt, ngt, None, fused=True)       # obs moves on while the gathers are in flight
        assert g.in_flight() == k
        assert len({o.data_ptr() for o in outs}) == k    # k distinct slots
        g.wait()
        assert g.in_flight() == 0
        for j in range(k):
            assert torch.equal(outs[j][:E], wants[j]), (rnd, j)


def _stepping_env(E=4096, V=8, M=64):
    env = make_vec(E, V, M, seed=3, yaml=True)
    env.make_new_game(); env.renew_positions(); env.compute_parms(); env.Random_phase()
    rng = np.random.default_rng(0)
    a = torch.from_numpy(rng.uniform(0, 1, (E, 2, V)).astype(np.float32)).cuda()
    pt = torch.full((E, V), -1, dtype=torch.int32, device="cuda:0")
    ngt = torch.full((E,), V, dtype=torch.int32, device="cuda:0")
    return env, a, pt, ngt


def test_joint_obs_gather_on_device_single_process():
    from ris_vec_marl_amd import dist as rdist
    env, a, pt, ngt = _stepping_env()
    g = rdist.JointObsGather(env.n_envs, env.n_veh, env.device)
    assert g.stream is not None and not g.collective and g.world == 1
    _check_gather(g, env, a, pt, ngt)
    with pytest.raises(ValueError):
        g.start(env.tensors["obs"][:10])
    _check_gathers_in_flight(rdist.JointObsGather(env.n_envs, env.n_veh, env.device, n_buffers=3), env, a, pt, ngt)
    # from a non-default main stream as well
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        env2, a2, pt2, ngt2 = _stepping_env(E=1024)
        g2 = rdist.JointObsGather(env2.n_envs, env2.n_veh, env2.device)
        _check_gather(g2, env2, a2, pt2, ngt2)
    torch.cuda.synchronize()


def test_joint_obs_gather_on_device_rccl_one_rank():
    """The same through a real RCCL communicator: a one-rank `nccl` process group on cuda:0 (the collective is
    issued on the side stream from the staged snapshot, exactly as with 8 ranks)."""
    import torch.distributed as td
    from ris_vec_marl_amd import dist as rdist
    if td.is_initialized():
        pytest.skip("a process group already exists in this process")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    torch.cuda.set_device(0)
    td.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    try:
        env, a, pt, ngt = _stepping_env()
        g = rdist.JointObsGather(env.n_envs, env.n_veh, env.device)
        assert g.collective and g.backend == "nccl" and g.world == 1
        _check_gather(g, env, a, pt, ngt)
        # three RCCL gathers in flight before the first wait (round 3: one slot per gather in flight)
        _check_gathers_in_flight(rdist.JointObsGather(env.n_envs, env.n_veh, env.device, n_buffers=3), env, a, pt, ngt)
        joint = rdist.gather_joint_obs(env.tensors["obs"])
        assert torch.equal(joint, env.tensors["obs"].reshape(env.n_envs, -1))
        with pytest.raises(ValueError):
            rdist.JointObsGather(8, 8, "cpu")            # RCCL cannot gather host tensors
    finally:
        td.destroy_process_group()


# ---------------------------------------------------------------------------- default device
def test_objects_built_with_the_default_device_interoperate():
    """Every constructor defaults to device="cuda"; torch compares `cuda` != `cuda:0`, so the device checks of
    the bound launchers must see the resolved device.  Two rollout steps through every bind_* launcher."""
    from ris_vec_marl_amd import (BatchedPolicy, EpisodeMeter, NomaGrouper, VecEnviron, VecReplayBuffer, apply_yaml_config,
                                  reference_lanes)
    L = reference_lanes()
    E, V, M = 256, 8, 36
    env = VecEnviron(L["down_lanes"], L["up_lanes"], L["left_lanes"], L["right_lanes"], 400, 400, V, M, 3, n_envs=E, seed=1)
    apply_yaml_config(env, None)
    env.make_new_game(); env.compute_parms(); env.Random_phase(); env.update_channel_gains()
    assert env.device == env.tensors["obs"].device and env.device.index is not None
    grouper = NomaGrouper(env)
    replay = VecReplayBuffer(8 * E, 5, V + 2, V)
    policy = BatchedPolicy(V, 5, 64, 32)
    meter = EpisodeMeter(env)
    dev = env.device
    action = torch.zeros(E, 2, V, device="cuda")
    p01 = torch.zeros(E, V, device="cuda")
    a_store = torch.zeros(E, V * (V + 2), device="cuda")
    grouper.begin_episode(0)
    mask = grouper.refresh_mask()
    partner, ng = grouper.group(p01, 0)
    assert partner.device == dev
    step = env.bind_step(action, partner, ng, None, fused=True)
    group = grouper.bind_group(p01)
    store = replay.bind_store(None, a_store, env.tensors["metrics"], env.tensors["reward"], env.tensors["obs"], mask)
    count = meter.bind(env)
    meter.begin_episode()
    for st in range(2):
        policy.choose_action(env.observe(), mask, cpu_share_floor=float(env.cpu_share_floor), want_onehot=False,
                             out=(action, p01, a_store))
        group(st)                       # second step: prev_global = env metrics (the check ADVICE flagged)
        step()
        store(done=False, use_mask=True)
        count()
    grouper.group(p01, 2)               # the unbound form takes the same path
    torch.cuda.synchronize()
    assert replay.mem_cntr == 2 * E and torch.isfinite(env.tensors["reward"]).all()


def test_observe_after_reset_and_checkpoint_guards():
    """observe() after make_new_game() reflects the new DataBuf (marl_get_state reads live attributes,
    marl_train_bcd.py:819-827); a checkpoint refuses an env with another seed / shard offset and carries power_w."""
    E, V, M = 64, 8, 16
    env = make_vec(E, V, M, seed=5, yaml=True)
    env.make_new_game(); env.compute_parms(); env.Random_phase()
    t = env.tensors
    o0 = env.observe().clone()
    assert torch.equal(o0[..., 0], t["data_buf"] / 10) and bool((o0[..., 1:] == 0).all())
    a = torch.rand(E, 2, V, device="cuda:0")
    pt = torch.full((E, V), -1, dtype=torch.int32, device="cuda:0"); ngt = torch.full((E,), V, dtype=torch.int32, device="cuda:0")
    env.step(a, pt, ngt, None, fused=True)
    o1 = env.observe().clone()
    env.make_new_game()
    o2 = env.observe()
    assert torch.equal(o2[..., 0], t["data_buf"] / 10)                        # the new backlog
    assert torch.equal(o2[..., 1:], o1[..., 1:])                              # data_t / data_p / rate survive a reset
    sd = env.state_dict()
    assert "power_w" in sd
    for kw in (dict(seed=6), dict(env_offset=64)):
        other = make_vec(E, V, M, **{**dict(seed=5), **kw}, yaml=True)
        with pytest.raises(ValueError):
            other.load_state_dict(sd)
    twin = make_vec(E, V, M, seed=5, yaml=True)
    twin.load_state_dict(sd)
    assert torch.equal(twin.tensors["power_w"], t["power_w"]) and torch.equal(twin.observe(), env.observe())


# ---------------------------------------------------------------------------- small batches / the T-step launch
def _rollout_env(E, V, M, seed=9):
    env = make_vec(E, V, M, seed=seed, yaml=True)
    env.make_new_game(); env.renew_positions(); env.compute_parms(); env.Random_phase()
    return env


@pytest.mark.parametrize("E,V,M", [(4096, 8, 36), (8192, 8, 64), (1000, 8, 40), (515, 4, 16), (1, 8, 64), (3, 8, 36),
                                   (301, 16, 256), (2050, 16, 64), (5, 16, 64),
                                   # every envs-per-wavefront instantiation of every shape (1 / 2 / 4 by batch size)
                                   (8192, 16, 64), (4100, 16, 64), (9000, 8, 40), (4500, 8, 40), (20000, 4, 16), (5000, 4, 16),
                                   (2100, 8, 64), (5000, 8, 64), (9000, 8, 36)])
def test_small_batch_kernel_is_the_pipelined_kernel_bit_for_bit(E, V, M):
    """Below ~12k envs `risvec_step_fused` takes the latency-shaped single-group kernel (k_step_lat.hip); same
    arithmetic in the same order as the software pipeline, so every output must be identical.  RISVEC_LAT_MAX_ENVS=0
    in a child process forces the pipeline for the comparison (the switch is read once per process)."""
    import subprocess
    import sys
    import tempfile
    rng = np.random.default_rng(E + M)
    action, partner, ng, arrivals = random_step_inputs(E, V, rng)
    env = _rollout_env(E, V, M)
    for _ in range(3):
        env.step(action.astype(np.float32), partner.astype(np.int32), ng.astype(np.int32), arrivals.astype(np.int32), fused=True)
    keys = ("gain", "reward", "data_buf", "mec_q", "rate", "data_t", "data_p", "over_power", "obs", "metrics", "power_w")
    mine = {k: cpu(env.tensors[k]).copy() for k in keys}
    with tempfile.TemporaryDirectory() as tmp:
        np.savez(os.path.join(tmp, "in.npz"), action=action, partner=partner, ng=ng, arrivals=arrivals)
        code = (
            "import sys, numpy as np; sys.path.insert(0, %r)\n"
            "from tests.test_entry_points_hip import _rollout_env, cpu\n"
            "z = np.load(%r)\n"
            "env = _rollout_env(%d, %d, %d)\n"
            "for _ in range(3):\n"
            "    env.step(z['action'].astype(np.float32), z['partner'].astype(np.int32), z['ng'].astype(np.int32), z['arrivals'].astype(np.int32), fused=True)\n"
            "np.savez(%r, **{k: cpu(env.tensors[k]) for k in %r})\n"
        ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(tmp, "in.npz"), E, V, M,
             os.path.join(tmp, "out.npz"), keys)
        e = dict(os.environ, RISVEC_LAT_MAX_ENVS="0")
        r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        other = np.load(os.path.join(tmp, "out.npz"))
        for k in keys:
            assert np.array_equal(mine[k], other[k]), k


@pytest.mark.parametrize("E,V,M,T", [(4096, 8, 36, 7), (8192, 8, 64, 4), (301, 8, 40, 5), (77, 4, 16, 3), (40000, 8, 64, 3),
                                     (130, 5, 21, 4), (64, 16, 256, 2), (9, 8, 64, 1), (3000, 8, 36, 3), (2500, 8, 40, 2),
                                     (9000, 4, 16, 3)])
@pytest.mark.parametrize("inject", [False, True])
def test_multi_step_launch_equals_single_launches(E, V, M, T, inject):
    """risvec_step_fused_multi == T consecutive risvec_step_fused calls, bit for bit: final state and outputs, and
    every step's trajectory record equals what the env's tensors held after that single step (compile-time shapes
    take the one-launch kernel; (5,21) and (16,256) take one fused launch for step 0 and the cached-gain multi-step
    kernel for the rest)."""
    rng = np.random.default_rng(E + M + T)
    _, partner, ng, _ = random_step_inputs(E, V, rng)
    actions = rng.uniform(-0.1, 1.2, (T, E, 2, V)).astype(np.float32)
    arrivals = rng.poisson(1.0, (T, E, V)).astype(np.int32) if inject else None
    pt, ngt = partner.astype(np.int32), ng.astype(np.int32)
    keys = ("gain", "reward", "data_buf", "mec_q", "rate", "data_t", "data_p", "over_power", "obs", "metrics")
    one = _rollout_env(E, V, M)
    per_step = []
    for t in range(T):
        one.step(actions[t], pt, ngt, None if arrivals is None else arrivals[t], fused=True, power_w=False)
        per_step.append({k: cpu(one.tensors[k]).copy() for k in ("reward", "obs", "metrics")})
    many = _rollout_env(E, V, M)
    rec = many.step_many(actions, pt, ngt, arrivals)
    assert many._steps == one._steps == T
    for k in keys:
        assert np.array_equal(cpu(many.tensors[k]), cpu(one.tensors[k])), k
    for t in range(T):
        for k in ("reward", "obs", "metrics"):
            assert np.array_equal(cpu(rec[k][t]), per_step[t][k]), (t, k)
    # and the launch after it continues the same Philox stream / state
    one.step(actions[0], pt, ngt, None, fused=True, power_w=False)
    many.step(actions[0], pt, ngt, None, fused=True, power_w=False)
    assert torch.equal(many.tensors["data_buf"], one.tensors["data_buf"])


@pytest.mark.parametrize("E,V,M,T", [(4096, 8, 36, 6), (301, 8, 40, 5), (130, 5, 21, 4), (257, 16, 256, 5), (77, 4, 16, 3),
                                     (1000, 13, 20, 3), (9, 8, 64, 1), (40000, 8, 64, 3)])
@pytest.mark.parametrize("inject", [False, True])
def test_cached_multi_step_launch_equals_single_launches(E, V, M, T, inject):
    """risvec_step_multi (`step_many(..., fused=False)`) == T consecutive risvec_step calls on the cached gains, bit for
    bit, for any shape: final state, outputs and every step's trajectory record -- the reference driver's own cadence
    (gains every 100 steps, step() every step; marl_train_bcd.py:1304-1611) in one launch."""
    rng = np.random.default_rng(E + M + T + 1)
    _, partner, ng, _ = random_step_inputs(E, V, rng)
    actions = rng.uniform(-0.1, 1.2, (T, E, 2, V)).astype(np.float32)
    arrivals = rng.poisson(1.0, (T, E, V)).astype(np.int32) if inject else None
    pt, ngt = partner.astype(np.int32), ng.astype(np.int32)
    keys = ("gain", "reward", "data_buf", "mec_q", "rate", "data_t", "data_p", "over_power", "obs", "metrics")
    one, many = _rollout_env(E, V, M), _rollout_env(E, V, M)
    one.update_channel_gains(); many.update_channel_gains()
    per_step = []
    for t in range(T):
        one.step(actions[t], pt, ngt, None if arrivals is None else arrivals[t], fused=False, power_w=False)
        per_step.append({k: cpu(one.tensors[k]).copy() for k in ("reward", "obs", "metrics")})
    rec = many.step_many(actions, pt, ngt, arrivals, fused=False)
    assert many._steps == one._steps == T
    for k in keys:
        assert np.array_equal(cpu(many.tensors[k]), cpu(one.tensors[k])), k
    for t in range(T):
        for k in ("reward", "obs", "metrics"):
            assert np.array_equal(cpu(rec[k][t]), per_step[t][k]), (t, k)
    # the bound launcher does the same, and the Philox stream / state continue
    launch = many.bind_step_many(torch.as_tensor(actions).cuda(), torch.as_tensor(pt).cuda(), torch.as_tensor(ngt).cuda(),
                                 None if arrivals is None else torch.as_tensor(arrivals).cuda(), fused=False)
    launch()
    for t in range(T):
        one.step(actions[t], pt, ngt, None if arrivals is None else arrivals[t], fused=False, power_w=False)
    for k in keys:
        assert np.array_equal(cpu(many.tensors[k]), cpu(one.tensors[k])), k


def test_multi_step_launch_options_and_errors():
    E, V, M, T = 300, 8, 36, 3
    rng = np.random.default_rng(0)
    _, partner, ng, _ = random_step_inputs(E, V, rng)
    pt, ngt = partner.astype(np.int32), ng.astype(np.int32)
    pol = rng.uniform(-1.2, 1.2, (T, E, V, 2)).astype(np.float32)
    a, b = _rollout_env(E, V, M), _rollout_env(E, V, M)
    for t in range(T):
        a.step(pol[t], pt, ngt, None, fused=True, policy_action=True)
    out = {"reward": torch.empty(T, E, V, device="cuda:0")}
    rec = b.step_many(pol, pt, ngt, None, policy_action=True, record=("reward",), out=out)
    assert rec["reward"] is out["reward"] and set(rec) == {"reward"}
    for k in ("reward", "data_buf", "obs", "metrics", "mec_q"):
        assert torch.equal(a.tensors[k], b.tensors[k]), k
    with pytest.raises(ValueError):
        b.step_many(pol[:, :, :, :1], pt, ngt)
    with pytest.raises(ValueError):
        b.step_many(pol, pt, ngt, record=("rewards",))
    with pytest.raises(ValueError):
        b.step_many(np.zeros((0, E, 2, V), np.float32), pt, ngt)
    import ctypes as C
    from ris_vec_marl_amd import _native as N
    lib = N.load()
    act = torch.zeros(T, E, 2, V, device="cuda:0")
    args = (C.byref(b._cstate), C.byref(b._p()), T, act.data_ptr(), torch.as_tensor(pt).cuda().data_ptr(),
            torch.as_tensor(ngt).cuda().data_ptr(), None, 0, 0)
    assert lib.risvec_step_fused_multi(*args, N.STEP_STEER, None, None) == N.ERR_ARG
    assert lib.risvec_step_fused_multi(*args[:2], 0, *args[3:], 0, None, None) == N.ERR_ARG


# ---------------------------------------------------------------------------- facade: reference semantics of odd group lists
def test_facade_takes_the_group_lists_the_reference_takes():
    """`Environ.step(action, noma_groups)` with a vehicle in several groups, pairs [u, u], groups of 3+ and empty
    groups against the reference's own outputs (tests/golden/facade_groups_8.npz, Environment.py:339-369), and the
    returned arrays are copies (the reference's are live aliases: tests/golden/facade_alias_8.npz)."""
    from ris_vec_marl_amd import Environ, reference_lanes
    from tests.test_hip_parity import load, set_params
    g = load("facade_groups_8.npz")
    n, V = g["gain"].shape
    L = reference_lanes()
    env = Environ(L["down_lanes"], L["up_lanes"], L["left_lanes"], L["right_lanes"], 400, 400, V, 16, 3, device="cuda:0")
    env.make_new_game()
    p = orc.OracleParams.yaml_effective()
    set_params(env, p)
    worst = 0.0
    n_checked = 0
    for i in range(0, n, 2):
        groups = [[int(u) for u in g["groups"][i, k, :m]] for k, m in enumerate(g["group_len"][i]) if m >= 0]
        env.DataBuf = g["data_buf0"][i]; env.mec_queue_cycles = g["mec_q0"][i]; env.channel_gains = g["gain"][i]
        r = env.step(g["action"][i], groups, arrivals=g["arrivals"][i])
        # the float32 image of the inputs through the oracle: masks for threshold proximity
        partner, ng = orc.encode_groups(groups, V)
        o = orc.step(g["data_buf0"][i][None].astype(np.float32).astype(np.float64), np.array([np.float32(g["mec_q0"][i])], dtype=np.float64),
                     g["gain"][i][None].astype(np.float32).astype(np.float64), g["action"][i][None], partner[None], np.array([ng]),
                     g["arrivals"][i][None], p)
        near_qos, near_other = step_mask(o, partner[None], g["gain"][i][None], np.array([g["mec_q0"][i]]))
        ok = ~near_other[0]
        rate = np.asarray(env.vehicle_rate)
        assert (np.abs(rate - g["vehicle_rate"][i]) <= RT * g["vehicle_rate"][i] + 1e-7)[ok].all(), i
        okr = ok & ~near_qos[0]
        d_scale = g["data_buf0"][i] * 1000 * p.cycles_per_bit / (p.cpu_share_floor * p.f_local_max)
        err = np.abs(np.asarray(r[0]) - g["reward"][i])
        # inputs were rounded to float32 on upload: |oracle(float32 inputs) - reference| is that rounding, evaluated exactly
        tol = RT * np.abs(g["reward"][i]) + 4e-7 * p.w_d * d_scale + 1e-9 + np.abs(o["reward"][0] - g["reward"][i])
        assert (err <= tol)[okr].all(), i
        if okr.any():
            worst = max(worst, float((err / np.abs(g["reward"][i]))[okr].max()))
        n_checked += int(okr.sum())
    assert n_checked > 0.9 * (n // 2) * V
    print("facade odd group lists: max reward rel err vs the reference %.2e" % worst)
    # copies, not aliases
    a = np.asarray(r[3]).copy()
    env.step(g["action"][0], [[0], [1]], arrivals=g["arrivals"][0])
    assert np.array_equal(np.asarray(r[3]), a)


# ---------------------------------------------------------------------------- non-temporal load variants
@pytest.mark.parametrize("E,V,M", [(2100, 8, 64), (300, 16, 256), (1500, 8, 36), (1500, 8, 80), (700, 16, 100), (900, 4, 24),
                                   (400, 8, 200)])
def test_non_temporal_variants_are_bit_identical(E, V, M):
    """Above ~270 MiB per step the software pipeline (MARL / SARL / gain cores) and k_colsum_slab read h_r / theta with
    the non-temporal hint; the hint must not change a bit.  The switch is read once per process, so a child process
    forces it on at a small size (RISVEC_PIPE_NT=1, RISVEC_COLSUM_NT=1, RISVEC_LAT_MAX_ENVS=0 to stay in the pipeline)."""
    import subprocess
    import sys
    import tempfile
    rng = np.random.default_rng(E + M)
    action, partner, ng, arrivals = random_step_inputs(E, V, rng)
    phase = rng.uniform(0, 2 * np.pi, (E, M)).astype(np.float32)
    code = (
        "import sys, numpy as np, torch; sys.path.insert(0, %r)\n"
        "from tests.test_entry_points_hip import _rollout_env, cpu\n"
        "z = np.load(sys.argv[1])\n"
        "env = _rollout_env(%d, %d, %d)\n"
        "out = {}\n"
        "env.rebuild_colsum(); out['c_col'] = cpu(env.tensors['c_col'])\n"
        "env.compute_parms(); env.optimize_phase_shift(); env.update_channel_gains(); out['gain_only'] = cpu(env.tensors['gain'])\n"
        "for _ in range(2):\n"
        "    env.step(z['action'].astype(np.float32), z['partner'].astype(np.int32), z['ng'].astype(np.int32), z['arrivals'].astype(np.int32), fused=True)\n"
        "for k in ('gain', 'reward', 'data_buf', 'metrics', 'obs', 'theta'): out[k] = cpu(env.tensors[k])\n"
        "env.sarl_step(np.clip(z['action'], 0, 1).astype(np.float32), z['phase'], z['arrivals'].astype(np.int32))\n"
        "for k in ('gain', 'reward', 'data_buf'): out['sarl_' + k] = cpu(env.tensors[k])\n"
        "np.savez(sys.argv[2], **out)\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), E, V, M)
    with tempfile.TemporaryDirectory() as tmp:
        np.savez(os.path.join(tmp, "in.npz"), action=action, partner=partner, ng=ng, arrivals=arrivals, phase=phase)
        outs = []
        for name, extra in (("default", dict(RISVEC_PIPE_NT="0", RISVEC_COLSUM_NT="0")), ("nt", dict(RISVEC_PIPE_NT="1", RISVEC_COLSUM_NT="1"))):
            e = dict(os.environ, RISVEC_LAT_MAX_ENVS="0", **extra)
            dst = os.path.join(tmp, name + ".npz")
            r = subprocess.run([sys.executable, "-c", code, os.path.join(tmp, "in.npz"), dst], env=e, capture_output=True, text=True,
                               timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            outs.append(np.load(dst))
        assert set(outs[0].files) == set(outs[1].files) and len(outs[0].files) == 11
        for k in outs[0].files:
            assert np.array_equal(outs[0][k], outs[1][k]), k
        # the latency-shaped kernel's non-temporal form (what risvec_step_fused takes beyond ~270 MB per step), forced
        # at this size with RISVEC_LAT_NT=1: same bits again (shapes without that kernel stay in the pipeline)
        e = dict(os.environ, RISVEC_LAT_NT="1", RISVEC_PIPE_NT="0", RISVEC_COLSUM_NT="0")
        dst = os.path.join(tmp, "lat_nt.npz")
        r = subprocess.run([sys.executable, "-c", code, os.path.join(tmp, "in.npz"), dst], env=e, capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lat = np.load(dst)
        for k in outs[0].files:
            assert np.array_equal(outs[0][k], lat[k]), k
        # and the form taken between 1 x and 1.29 x the Infinity Cache: default cache policy, the envs walked in alternating
        # directions from step to step (RISVEC_LAT_PINGPONG=1 forces it at this size)
        e = dict(os.environ, RISVEC_LAT_PINGPONG="1", RISVEC_PIPE_NT="0", RISVEC_COLSUM_NT="0")
        dst = os.path.join(tmp, "lat_alt.npz")
        r = subprocess.run([sys.executable, "-c", code, os.path.join(tmp, "in.npz"), dst], env=e, capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        alt = np.load(dst)
        for k in outs[0].files:
            assert np.array_equal(outs[0][k], alt[k]), k


# ---------------------------------------------------------------------------- run-time-M members of the fused family
@pytest.mark.parametrize("E,V,M", [(3001, 8, 20), (3001, 8, 120), (2049, 8, 256), (4100, 4, 100), (1027, 16, 50), (515, 16, 120)])
def test_runtime_m_kernel_forms_agree(E, V, M):
    """A run-time-M shape takes k_step_fused_lat<V, M=.. (G, NIT), EPWT> with 1 / 2 / 4 envs per wavefront by batch size
    (and the non-temporal form beyond the Infinity Cache): every form must produce the same bits.  RISVEC_LAT_EPW /
    RISVEC_LAT_NT are read once per process, so child processes force each form; the kernel name each child reports
    is asserted so that a dispatch change cannot silently untest a form."""
    import subprocess
    import sys
    import tempfile
    rng = np.random.default_rng(E + M)
    action, partner, ng, arrivals = random_step_inputs(E, V, rng)
    code = (
        "import sys, numpy as np, torch; sys.path.insert(0, %r)\n"
        "from tests.test_entry_points_hip import _rollout_env, cpu\n"
        "from ris_vec_marl_amd import _native as N\n"
        "z = np.load(sys.argv[1])\n"
        "env = _rollout_env(%d, %d, %d)\n"
        "for _ in range(3):\n"
        "    env.step(z['action'].astype(np.float32), z['partner'].astype(np.int32), z['ng'].astype(np.int32), z['arrivals'].astype(np.int32), fused=True)\n"
        "out = {k: cpu(env.tensors[k]) for k in ('gain', 'reward', 'data_buf', 'mec_q', 'rate', 'metrics', 'obs', 'power_w')}\n"
        "out['kernel'] = np.array(N.last_kernel())\n"
        "np.savez(sys.argv[2], **out)\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), E, V, M)
    forms = [("epw1", dict(RISVEC_LAT_EPW="1")), ("epw2", dict(RISVEC_LAT_EPW="2")), ("epw4", dict(RISVEC_LAT_EPW="4")),
             ("nt", dict(RISVEC_LAT_NT="1"))]
    with tempfile.TemporaryDirectory() as tmp:
        np.savez(os.path.join(tmp, "in.npz"), action=action, partner=partner, ng=ng, arrivals=arrivals)
        outs, names = [], []
        for name, extra in forms:
            dst = os.path.join(tmp, name + ".npz")
            r = subprocess.run([sys.executable, "-c", code, os.path.join(tmp, "in.npz"), dst], env=dict(os.environ, **extra),
                               capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            outs.append(np.load(dst))
            names.append(str(outs[-1]["kernel"]))
        assert all(n.startswith("k_step_fused_lat<%d,M=%d" % (V, M)) for n in names), names
        assert names[3].endswith(",NT>") and not names[1].endswith(",NT>"), names
        assert len(set(names)) >= 3, names                     # at least three distinct members ran
        for o in outs[1:]:
            for k in outs[0].files:
                if k != "kernel":
                    assert np.array_equal(outs[0][k], o[k]), k


# ---------------------------------------------------------------------------- theta kept by index (lazy_theta)
@pytest.mark.parametrize("V,M,E", [(16, 256, 96), (8, 64, 300), (8, 64, 5000), (8, 36, 300), (8, 100, 257), (4, 24, 200),
                                   (16, 64, 150), (16, 20, 100), (8, 250, 64)])
def test_theta_by_index_is_bit_identical(V, M, E):
    """`lazy_theta=True`: between BCD sweeps theta lives as one candidate index per element -- `step(bcd=True)` takes the
    sweep that does not write the complex64 tensor (RISVEC_STEP_THETA_BY_INDEX) and the fused step kernel that expands
    the indices (k_step_fused_lat<.., TK>).  Every output of every step, and the tensor itself whenever it is asked
    for, must equal the default mode's bit for bit -- through phase setters, plain fused steps, a gain refresh, a
    stand-alone sweep and a checkpoint."""
    from ris_vec_marl_amd import _native as N
    rng = np.random.default_rng(V * M + E)
    action, partner, ng, _ = random_step_inputs(E, V, rng)
    a, pt, ngt = action.astype(np.float32), partner.astype(np.int32), ng.astype(np.int32)
    envs = []
    for lazy in (False, True):
        env = _rollout_env(E, V, M)
        env.lazy_theta = lazy
        envs.append(env)
    ref, lz = envs
    keys = ("gain", "reward", "data_buf", "mec_q", "rate", "data_t", "data_p", "metrics", "obs", "power_w")

    def same(what, theta=False):
        for k in keys + (("theta", "theta_idx", "s_sum") if theta else ()):
            assert np.array_equal(cpu(ref.tensors[k]), cpu(lz.tensors[k])), (what, k)

    n_by_index = 0
    for i in range(5):
        arr = rng.poisson(1.0, (E, V)).astype(np.int32)
        for env in envs:
            env.step(a, pt, ngt, arr, fused=True, bcd=True)
        if i >= 1:                                             # the first sweep (indices unknown) writes theta in both modes
            assert lz._theta_stale and ",TK" in N.last_kernel(), N.last_kernel()
            n_by_index += 1
        assert not ref._theta_stale
        same("bcd step %d" % i, theta=(i % 2 == 1))            # asking for tensors materialises theta
    assert n_by_index == 4
    for env in envs:                                           # a plain fused step on the swept theta: by index too
        env.step(a, pt, ngt, None, fused=True)
    assert ",TK" in N.last_kernel()
    for env in envs:
        env.step(a, pt, ngt, None, fused=True, bcd=True)
        env.update_channel_gains()                             # a consumer of the complex64 tensor: materialises it first
    assert not lz._theta_stale
    same("gain refresh", theta=True)
    for env in envs:
        env.optimize_phase_shift()                             # stand-alone sweep: indices only in lazy mode
    assert lz._theta_stale
    same("stand-alone sweep", theta=True)
    for env in envs:
        env.Random_phase()                                     # a setter makes the tensor the truth again
        env.step(a, pt, ngt, None, fused=True, bcd=True)       # generic sweep (indices unknown)
        env.step(a, pt, ngt, None, fused=True, bcd=True)
    same("after Random_phase", theta=True)
    # bound launcher + checkpoint round trip
    a_dev, pt_dev, ng_dev = torch.from_numpy(a).cuda(), torch.from_numpy(pt).cuda(), torch.from_numpy(ngt).cuda()
    launches = [env.bind_step(a_dev, pt_dev, ng_dev, None, fused=True, bcd=True) for env in envs]
    for _ in range(3):
        for launch in launches:
            launch()
    assert lz._theta_stale
    sd = lz.state_dict()
    assert not lz._theta_stale
    assert np.array_equal(sd["theta"].numpy(), cpu(ref.tensors["theta"]))
    same("bound launches", theta=True)


# ---------------------------------------------------------------------------- full-size tests of what the big configs dispatch
def test_full_size_properties_c5():
    """BASELINE configs[4] (32 768 x 16 x 256, `step(bcd=True)` every step) through the kernels that configuration is
    benchmarked on -- k_bcd_sweep8_pair (theta by index) + k_step_fused_lat<16,256,1,NT,TK> -- with size-independent
    checks: which kernel ran (a threshold change must not silently untest them); whole batch == two half batches with
    theta written eagerly, bit for bit (shard independence, lazy == eager theta, RNG keyed by global env id); the BCD
    objective never decreases; kbit conservation; a 128-env shard (default cache policy, eager theta) equals the same
    envs of the big batch and tracks the oracle's sweep -> gains -> step; checkpoint round trip.
    Reference: Environment.py:208-231, 255-273, 547-731."""
    from ris_vec_marl_amd import _native as N
    E, V, M = 32768, 16, 256
    rng = np.random.default_rng(5)
    action = torch.from_numpy(rng.uniform(0, 1, (E, 2, V)).astype(np.float32)).cuda()
    pnp, gnp = np.full((E, V), -1, dtype=np.int32), np.full(E, V - 2, dtype=np.int32)
    pnp[:, 0], pnp[:, 1], pnp[:, 4], pnp[:, 9] = 1, 0 + (1 << 16), 9, 4 + (1 << 16)
    partner, ng = torch.from_numpy(pnp).cuda(), torch.from_numpy(gnp).cuda()

    def build(n, lo, lazy):
        env = make_vec(n, V, M, seed=3, env_offset=lo, yaml=True)
        env.lazy_theta = lazy
        env.make_new_game()
        for _ in range(2):
            env.renew_positions()
        env.compute_parms()
        env.Random_phase()
        return env

    def objective(env):
        t = env.tensors
        th = torch.view_as_complex(t["theta"]); hr = torch.view_as_complex(t["h_r"]); b = torch.view_as_complex(t["b"])
        return (th * hr.sum(1) * b[None]).sum(1).abs().double() ** 2

    keys = ("reward", "gain", "data_buf", "mec_q", "rate", "metrics", "obs", "theta")
    whole = build(E, 0, True)
    B0 = whole.tensors["data_buf"].clone()
    obj = [objective(whole)]
    names = []
    for i in range(3):
        whole.step(action, partner, ng, None, fused=True, bcd=True)
        names.append(N.last_kernel())
        obj.append(objective(whole))                     # (materialises theta: the next step is by index again)
    assert names[0] == "k_step_fused_lat<16,256,1,NT>", names          # first sweep: indices unknown, theta written
    assert names[1] == names[2] == "k_step_fused_lat<16,256,1,NT,TK>", names
    for a, b in zip(obj[:-1], obj[1:]):
        assert bool((b >= a * (1 - 1e-5)).all())         # coordinate ascent (the check itself is a complex64 sum)
    got = {k: whole.tensors[k].clone() for k in keys}
    # kbit conservation of the last step: what left the backlog was processed locally or offloaded
    m = got["metrics"]
    assert bool((m[:, 1] >= 0).all()) and bool((m[:, 2] >= 0).all())
    for lo, hi in ((0, E // 2), (E // 2, E)):
        h = build(hi - lo, lo, False)
        for i in range(3):
            h.step(action[lo:hi], partner[lo:hi], ng[lo:hi], None, fused=True, bcd=True)
        assert "NT" in N.last_kernel() and "TK" not in N.last_kernel()
        for k in keys:
            assert torch.equal(h.tensors[k], got[k][lo:hi]), k
        del h
        torch.cuda.empty_cache()
    # a small shard: default cache policy, eager theta -- the same bits, and the oracle's sweep -> gains -> step
    n = 128
    small = build(n, 0, False)
    p = orc.OracleParams.yaml_effective()
    for i in range(3):
        t = small.tensors
        th_in = snap(c128(t["theta"]), 3)
        Bq, Qq = cpu(t["data_buf"]).astype(np.float64), cpu(t["mec_q"]).astype(np.float64)
        arrivals = orc.philox_arrivals(np.arange(n), V, small._steps, 3, p.rate)
        o_th, o_idx, safe, h_, b_ = _bcd_step_reference(t, th_in, Bq, Qq, None, None, None, None, p, 3)
        out = small.step(action[:n], partner[:n], ng[:n], None, fused=True, bcd=True)
        assert N.last_kernel() == "k_step_fused_lat<16,256,1>"
        th1 = c128(t["theta"])
        record("BCD decisions left out of the comparison (fraction; float64 margin below 1e-9)", 1.0 - safe.mean())
        assert safe.mean() > 0.999 and np.abs(th1 - o_th)[safe].max() <= 1.5e-7
        g_dev = cpu(t["gain"]).astype(np.float64)
        o = orc.step(Bq, Qq, g_dev, cpu(action[:n]).astype(np.float64), pnp[:n], gnp[:n], arrivals, p)
        near_qos, near_other = step_mask(o, pnp[:n], g_dev, Qq)
        okr = check_step(small, out, o, Bq, p, near_qos, near_other)
        record("samples left out of the reward comparison (fraction; discontinuities of step())", 1.0 - okr.mean())
        assert okr.mean() > 0.999
    for k in keys:
        assert torch.equal(small.tensors[k], got[k][:n]), k
    # checkpoint round trip at full size (the indices are re-derived from the restored theta)
    sd = whole.state_dict()
    env2 = build(E, 0, True)
    env2.load_state_dict(sd)
    a = [x.clone() for x in whole.step(action, partner, ng, None, fused=True)]
    b2 = [x.clone() for x in env2.step(action, partner, ng, None, fused=True)]
    for x, y in zip(a, b2):
        assert torch.equal(x, y)
    del B0


def test_full_size_dispatch_beyond_the_infinity_cache():
    """262 144 x 8 x 64 (1.36 GB per step: bench.py's `hbm_only` leg) takes k_step_fused_lat<8,64,4,NT> by size; it must
    equal the software pipeline (forced in a child process, where it runs with non-temporal loads as well) bit for
    bit, and the whole batch must equal its two halves."""
    import subprocess
    import sys
    import tempfile
    from ris_vec_marl_amd import _native as N
    E, V, M = 262144, 8, 64
    code = (
        "import sys, numpy as np, torch; sys.path.insert(0, %r)\n"
        "from tests.test_entry_points_hip import _big_rollout\n"
        "from ris_vec_marl_amd import _native as N\n"
        "out = _big_rollout(%d, %d, %d)\n"
        "out['kernel'] = np.array(N.last_kernel())\n"
        "np.savez(sys.argv[1], **out)\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), E, V, M)
    mine = _big_rollout(E, V, M)
    assert N.last_kernel() == "k_step_fused_lat<8,64,4,NT>", N.last_kernel()
    with tempfile.TemporaryDirectory() as tmp:
        dst = os.path.join(tmp, "pipe.npz")
        r = subprocess.run([sys.executable, "-c", code, dst], env=dict(os.environ, RISVEC_LAT_MAX_ENVS="0"), capture_output=True,
                           text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        other = np.load(dst)
        assert str(other["kernel"]).startswith("k_step_fused_pipe<8,64,2,MarlCore,NT>"), str(other["kernel"])
        for k in mine:
            assert np.array_equal(mine[k], other[k]), k
    for lo, hi in ((0, E // 2), (E // 2, E)):
        half = _big_rollout(hi - lo, V, M, lo=lo)
        for k in mine:
            assert np.array_equal(half[k], mine[k][lo:hi]), k


def test_full_size_dispatch_just_beyond_the_infinity_cache():
    """65 536 x 8 x 64 (302 MB of h_r + theta per step, 1.13 x the Infinity Cache): k_step_fused_lat<8,64,4,ALT> -- default cache policy,
    envs walked in alternating directions from step to step -- must equal its two halves (32 768 envs each: the software
    pipeline) bit for bit over three steps (both walking directions)."""
    from ris_vec_marl_amd import _native as N
    E, V, M = 65536, 8, 64
    mine = _big_rollout(E, V, M, steps=3)
    assert N.last_kernel() == "k_step_fused_lat<8,64,4,ALT>", N.last_kernel()
    for lo, hi in ((0, E // 2), (E // 2, E)):
        half = _big_rollout(hi - lo, V, M, lo=lo, steps=3)
        assert N.last_kernel().startswith("k_step_fused_pipe<8,64,2,MarlCore"), N.last_kernel()
        for k in mine:
            assert np.array_equal(half[k], mine[k][lo:hi]), k


def _big_rollout(E, V, M, lo=0, steps=2):
    env = make_vec(E, V, M, seed=4, env_offset=lo, yaml=True)
    env.make_new_game(); env.renew_positions(); env.compute_parms(); env.Random_phase()
    rng = np.random.default_rng(77)
    action = rng.uniform(0, 1, (lo + E, 2, V)).astype(np.float32)[lo:]
    partner = np.full((E, V), -1, dtype=np.int32); partner[:, 2], partner[:, 5] = 5, 2 + (1 << 16)
    ng = np.full(E, V - 1, dtype=np.int32)
    a, pt, ngt = torch.from_numpy(action).cuda(), torch.from_numpy(partner).cuda(), torch.from_numpy(ng).cuda()
    for _ in range(steps):
        env.step(a, pt, ngt, None, fused=True)
    return {k: cpu(env.tensors[k]) for k in ("gain", "reward", "data_buf", "mec_q", "metrics", "obs")}
