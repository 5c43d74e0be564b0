"""GPU parity tests of the per-episode metrics sink: the device accumulators against
the episode scalars captured from the reference env with the driver's own bookkeeping statements
(`tests/golden/episode_metrics_*.npz`) and against oracle/metrics_oracle.py on a live vectorised env.

Tolerances: the sums are float64 on the device in the driver's order, so against the oracle fed the
SAME float32 step values the per-env scalars agree to 1e-13 relative; against the reference's float64
step values the only difference is the float32 rounding of each step's inputs (1e-6 relative)."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from oracle import metrics_oracle as MO  # noqa: E402  (checker)

GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"


def T(x, dt=np.float32):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=dt))).to(DEV)


def pad_metrics(m14):
    m = np.zeros(m14.shape[:-1] + (16,), dtype=np.float32)
    m[..., :14] = m14
    return m


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "episode_metrics_*.npz"))), ids=os.path.basename)
def test_episode_scalars_vs_golden(path):
    from ris_vec_marl_amd import EpisodeMeter
    d = np.load(path)
    V, n_env, n_ep, n_step = (int(x) for x in d["shape"])
    meter = EpisodeMeter(n_envs=n_env, n_veh=V, device=DEV)
    for ep in range(n_ep):
        meter.begin_episode()
        orcs = [MO.EpisodeOracle(V) for _ in range(n_env)]
        for st in range(n_step):
            m32, r32, p32 = pad_metrics(d["metrics"][:, ep, st]), d["reward"][:, ep, st].astype(np.float32), \
                d["power_w"][:, ep, st].astype(np.float32)
            meter.accumulate(metrics=T(m32), reward=T(r32), power_w=T(p32))
            for e in range(n_env):
                orcs[e].accumulate(m32[e, :14], r32[e], p32[e])
        out = meter.end_episode()
        got = meter.per_env.cpu().numpy()
        same_inputs = np.array([[o.end_episode()[c] for c in MO.COLUMNS] for o in orcs])
        np.testing.assert_allclose(got, same_inputs, rtol=1e-13, atol=1e-300)
        np.testing.assert_allclose(got, d["episode"][:, ep], rtol=2e-6, atol=1e-12)      # float32 step inputs
        s = meter.summary.cpu().numpy()
        np.testing.assert_allclose(s[0], got.mean(0), rtol=1e-13)
        assert np.array_equal(s[1], got.min(0)) and np.array_equal(s[2], got.max(0))
        assert out["reward/jain"] == s[0][19] and out["abs/delay_ms"] == 1000.0 * out["delay/episode_mean"]
        lo, hi = meter.spread()["reward/min_user"]
        assert lo == got[:, 17].min() and hi == got[:, 17].max()


def _make_env(E, V, M, seed):
    from ris_vec_marl_amd import VecEnviron, reference_lanes, apply_yaml_config
    L = reference_lanes()
    env = VecEnviron(L["down_lanes"], L["up_lanes"], L["left_lanes"], L["right_lanes"], 400, 400, V, M, 3,
                     n_envs=E, device=DEV, seed=seed)
    apply_yaml_config(env, None)
    env.make_new_game(); env.renew_positions(); env.compute_parms(); env.Random_phase(); env.update_channel_gains()
    return env


def test_live_env_episode_and_reduction_over_envs(tmp_path):
    """E = 3000 (not a multiple of the 256-env workgroup) for 2 episodes x 30 steps: every env's
    episode scalars against the oracle fed from the env's own tensors, the mean / min / max over the
    envs, a mid-episode checkpoint of the meter, and the sink's event file."""
    from ris_vec_marl_amd import EpisodeMeter, ScalarSink, encode_noma_groups
    from ris_vec_marl_amd.metrics import read_events, COLUMNS
    E, V, M = 3000, 8, 36
    env = _make_env(E, V, M, 5)
    meter = EpisodeMeter(env)
    launch = meter.bind(env)
    gen = torch.Generator(device=DEV).manual_seed(3)
    partner, ng = encode_noma_groups([[[0, 5], [1], [2, 7], [3], [4], [6]]], V)
    partner = torch.as_tensor(partner, dtype=torch.int32, device=DEV).repeat(E, 1).contiguous()
    ng = torch.full((E,), int(ng[0]), dtype=torch.int32, device=DEV)
    probe = [0, 1, 255, 256, 1777, E - 1]
    with ScalarSink(str(tmp_path)) as sink:
        for ep in range(2):
            meter.begin_episode()
            orcs = {e: MO.EpisodeOracle(V) for e in probe}
            snap = None
            for st in range(30):
                a = torch.rand(E, 2, V, device=DEV, generator=gen)
                env.step(a, partner, ng, None, fused=True)
                launch()
                if st == 11:
                    snap = meter.state_dict()
                m, r, p = (env.tensors[k].cpu().numpy() for k in ("metrics", "reward", "power_w"))
                for e in probe:
                    orcs[e].accumulate(m[e, :14], r[e], p[e])
            assert meter.n_steps == 30
            scalars = sink.write_episode(meter, ep, extra={"gumbel/tau": 1.0})
            got = meter.per_env.cpu().numpy()
            for e in probe:
                want = np.array([orcs[e].end_episode()[c] for c in MO.COLUMNS])
                np.testing.assert_allclose(got[e], want, rtol=1e-13, atol=1e-300, err_msg="env %d" % e)
            s = meter.summary.cpu().numpy()
            np.testing.assert_allclose(s[0], got.mean(0), rtol=1e-12)
            assert np.array_equal(s[1], got.min(0)) and np.array_equal(s[2], got.max(0))
            assert np.isfinite(got).all() and (got[:, 19] > 0).all() and (got[:, 19] <= 1 + 1e-12).all()
            assert scalars["reward/global_avg"] == s[0][0] and scalars["gumbel/tau"] == 1.0
            # a meter restored from the mid-episode snapshot continues to the same sums
            m2 = EpisodeMeter(env)
            m2.load_state_dict(snap)
            assert m2.n_steps == 12 and torch.equal(m2.acc, T(snap["acc"].numpy(), np.float64))
    ev = read_events(sink.path)
    assert {t for (_, _, t, _) in ev} >= set(COLUMNS) | {"abs/delay_ms", "power/total_W", "gumbel/tau"}
    assert sorted({s for (_, s, _, _) in ev}) == [0, 1]


def test_without_power_and_error_paths():
    from ris_vec_marl_amd import EpisodeMeter
    E, V = 70, 5
    meter = EpisodeMeter(n_envs=E, n_veh=V, device=DEV, user_clip=2.0)
    with pytest.raises(RuntimeError):
        meter.end_episode()
    rng = np.random.default_rng(0)
    m = pad_metrics(rng.normal(size=(E, 14)))
    r = rng.normal(0, 3, (E, V)).astype(np.float32)
    meter.accumulate(metrics=T(m), reward=T(r))                         # an env without last_power_W
    meter.accumulate(metrics=T(m), reward=T(r))
    meter.end_episode()
    got = meter.per_env.cpu().numpy()
    assert np.array_equal(got[:, 14:17], np.zeros((E, 3)))
    user = np.clip(r.astype(np.float64), -2, 2)
    np.testing.assert_allclose(got[:, 17], user.min(1), rtol=1e-15)
    np.testing.assert_allclose(got[:, 1], 2.0 * m[:, 1].astype(np.float64), rtol=1e-15)     # a SUM, not a mean
    assert np.array_equal(got[:, 3], m[:, 3].astype(np.float64)) and np.array_equal(got[:, 20], m[:, 0].astype(np.float64))
    with pytest.raises(ValueError):
        meter.accumulate(metrics=T(m[:, :14]), reward=T(r))
    with pytest.raises(ValueError):
        meter.accumulate(metrics=T(m), reward=T(r).double())
    with pytest.raises(ValueError):
        EpisodeMeter(n_envs=4, n_veh=65, device=DEV)


def test_full_size_accumulate_is_linear():
    """BASELINE C3 size (E = 32 768, V = 8): k accumulations of the same step equal k x one step
    (exactly: small integers times float32 values in float64), and the reduction over the envs
    matches numpy."""
    from ris_vec_marl_amd import EpisodeMeter
    E, V = 32768, 8
    env = _make_env(E, V, 64, 9)
    a = torch.rand(E, 2, V, device=DEV)
    from ris_vec_marl_amd import encode_noma_groups
    partner, ng = encode_noma_groups([[[0, 1], [2, 3], [4], [5], [6], [7]]], V)
    env.step(a, torch.as_tensor(partner, dtype=torch.int32, device=DEV).repeat(E, 1).contiguous(),
             torch.full((E,), int(ng[0]), dtype=torch.int32, device=DEV), None, fused=True)
    meter = EpisodeMeter(env)
    launch = meter.bind(env)
    launch()
    one = meter.acc.clone()
    for _ in range(7):
        launch()
    fixed = [c for c in range(meter.acc.shape[0]) if c != 16]
    assert torch.equal(meter.acc[fixed], 8.0 * one[fixed]) and torch.equal(meter.acc[16], one[16])
    meter.end_episode()
    got = meter.per_env.cpu().numpy()
    m = env.tensors["metrics"].cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(got[:, [0, 12, 13]], m[:, [0, 12, 13]], rtol=1e-15)
    np.testing.assert_allclose(meter.summary[0].cpu().numpy(), got.mean(0), rtol=1e-12)
