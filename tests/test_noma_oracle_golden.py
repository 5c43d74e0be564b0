"""oracle/noma_oracle.py against the vectors captured from the reference's own pairing code
(tools/capture_golden_noma.py): masks, pairs and counters exact, float64 thresholds / scores
bit-exact.  CPU only.

Tie policy (see the oracle's header): where the reference sorts EQUAL keys with np.argsort
(TRAIN:151, 271) its answer depends on the host's SIMD sort.  Cases without such ties must match
with the build's defined order (`stable=True`); cases with ties are compared with
`stable=False` -- np.argsort called the way the reference calls it -- and only when this host's
argsort reproduces a probe recorded at capture time."""
import glob
import os

import numpy as np
import pytest

from oracle import noma_oracle as NO

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name))


def groups_of(partner):
    pairs = sorted((i, int(p)) for i, p in enumerate(partner) if 0 <= p < (1 << 16))
    return pairs


def same_sort_as_capture_host():
    """Does np.argsort order equal keys here as it did where the fixtures were made?  (AVX-512
    hosts use a SIMD network; others a stable insertion sort for short rows.)"""
    a = -np.array([3., 1., 3., 0., 1., 3., 0., 2.])
    return not np.array_equal(np.argsort(a), np.argsort(a, kind="stable"))


HELPER_FILES = sorted(glob.glob(os.path.join(GOLD, "noma_helpers_*.npz")))
EPISODE_FILES = sorted(glob.glob(os.path.join(GOLD, "noma_episodes_*.npz")))


def test_fixtures_present():
    assert len(HELPER_FILES) == 3 and len(EPISODE_FILES) == 6


@pytest.mark.parametrize("path", HELPER_FILES, ids=os.path.basename)
def test_helpers(path):
    d = np.load(path)
    N = int(d["N"])
    prm = NO.NomaParams(noise_power=float(d["noise_power"]), P_max=float(d["P_max"]),
                        qos_R_min_bpsHz=float(d["R_min"]))
    simd = same_sort_as_capture_host()
    n_tie_checked = 0
    for c in range(len(d["gain"])):
        g = d["gain"][c]
        # dB gains: numpy on this host vs the capture (same library, normally identical); the
        # later stages take the captured values so that rounding-level ties resolve identically
        assert np.allclose(NO.gain_db(g, 1e-15), d["gdb15"][c], rtol=1e-15, atol=0)
        gdb15, gdb12 = d["gdb15"][c], d["gdb12"][c]
        tau = NO.adaptive_threshold(gdb15, float(d["q"][c]))
        assert tau == d["tau"][c]
        if not d["topk_tie"][c]:
            assert np.array_equal(NO.feasible_mask(gdb15, tau, int(d["K"][c])), d["mask"][c])
        mask = d["mask"][c]
        qos_in = None if d["qos"][c][0, 0] == 255 else d["qos"][c]
        assert np.array_equal(NO.qos_soft_mask(g, d["p01"][c].astype(float), prm), d["qos_ok"][c])
        S = NO.score_matrix(gdb12, mask, d["hist"][c], prm, qos_in)
        assert np.array_equal(S, d["S"][c])
        pairs = NO.mwm_primary(S, mask, float(d["accept_q"][c]), True)
        assert pairs == groups_of(d["mwm_partner"][c]), c
        pairs_ns = NO.mwm_primary(S, mask, float(d["accept_q"][c]), False)       # allow_singles=False quirks
        assert pairs_ns == groups_of(d["mwm_nosingles_partner"][c]), c
        prm2 = NO.NomaParams(score_w_delta_db=0.7, score_w_history=0.45, abs_gain_min_db=float(d["abs_min"][c]),
                             qos_soft_penalty_dbscore=2.5)
        assert np.array_equal(NO.score_matrix(gdb12, mask, d["hist"][c], prm2, qos_in), d["S_absmin"][c]), c
        comp = NO.mwm_completion(S, mask, pairs, int(d["min_pairs"][c]), 0.30)
        assert sorted(comp) == groups_of(d["comp_partner"][c]), c
        assert len(comp) == d["comp_npairs"][c]
        if not d["row_tie"][c]:
            rel = NO.relax_mask_once(mask, gdb12, float(d["relax_tau"][c]), int(d["relax_topk"][c]), stable=True)
            assert np.array_equal(rel, d["relax_mask"][c])
        elif simd:
            rel = NO.relax_mask_once(mask, gdb12, float(d["relax_tau"][c]), int(d["relax_topk"][c]), stable=False)
            assert np.array_equal(rel, d["relax_mask"][c])
            n_tie_checked += 1
        S2 = NO.score_matrix(gdb12, d["relax_mask"][c], d["hist"][c], prm, qos_in)
        assert np.array_equal(S2, d["S_relaxed"][c])


def params_of(d):
    cfg = dict(zip([str(k) for k in d["cfg_keys"]], d["cfg_vals"]))
    p = NO.NomaParams(noise_power=float(d["noise_power"]), P_max=float(d["P_max"]))
    for k, v in cfg.items():
        if hasattr(p, k):
            cur = getattr(p, k)
            setattr(p, k, bool(v) if isinstance(cur, bool) else int(v) if isinstance(cur, int) else float(v))
    return p


def policy_to_p01(policy):
    """TRAIN:1391-1396: float32 clip, (x + 1) / 2 in float32, stored as float64."""
    clipped = np.clip(policy[:, 0], np.float32(-0.999), np.float32(0.999))
    return ((clipped + np.float32(1)) / np.float32(2.0)).astype(np.float64)


@pytest.mark.parametrize("path", EPISODE_FILES, ids=os.path.basename)
def test_episodes(path):
    d = np.load(path)
    N = int(d["N"])
    prm = params_of(d)
    simd = same_sort_as_capture_host()
    n_ep, n_steps = d["gain"].shape[:2]
    checked = skipped = 0
    for e in range(n_ep):
        ep = NO.NomaEpisode(N)
        i_episode = int(d["first_episode"]) + e * 37
        diverged = False
        for t in range(n_steps):
            mask = None
            if d["has_mask"][e, t]:
                mask_o = NO.rebuild_mask(ep, d["gdb15"][e, t], prm, i_episode)
                assert ep.last_tau == d["tau_now"][e, t] and ep.last_K == d["K_now"][e, t]
                assert ep.last_q == d["q_now"][e, t]
                mask = d["mask"][e, t]
                if np.array_equal(mask_o, mask) is False:
                    # only a top-K tie at TRAIN:151 may differ; continue from the captured mask
                    gap = np.abs(d["gdb15"][e, t][:, None] - d["gdb15"][e, t][None, :])
                    assert any(len(set(r)) < N for r in gap)
            tie = bool(d["row_tie"][e, t])
            stable = not (tie and simd)
            groups, info = NO.group_step(ep, d["gain"][e, t], policy_to_p01(d["policy"][e, t]), mask, prm,
                                         i_episode, t, u_unstick=d["u_unstick"][e, t], stable=stable,
                                         gdb15=d["gdb15"][e, t], gdb12=d["gdb12"][e, t])
            partner, ng = NO.partner_of_groups(groups, N)
            exact_expected = (not tie) or simd or d["rounds"][e, t] == 0
            if exact_expected and not diverged:
                assert np.array_equal(partner, d["partner"][e, t]), (e, t)
                assert ng == d["n_groups"][e, t] and info["n_pairs"] == d["n_pairs"][e, t]
                assert info["recomputed"] == bool(d["recomputed"][e, t])
                if info["recomputed"]:
                    assert info["rounds"] == d["rounds"][e, t]
                assert np.array_equal(ep.hist, d["hist"][e, t])
                assert np.array_equal(ep.streak, d["streak"][e, t])
                assert ep.unstick_used == bool(d["unstick_used"][e, t])
                checked += 1
            else:
                diverged = diverged or not np.array_equal(partner, d["partner"][e, t])
                skipped += 1
            ep.observe_reward(float(d["global_reward"][e, t]))
    assert checked > 0.5 * n_ep * n_steps, (checked, skipped)


def test_quantile_matches_numpy():
    rng = np.random.default_rng(0)
    for _ in range(2000):
        n = int(rng.integers(1, 70))
        v = rng.normal(size=n)
        q = float(rng.choice([0.0, 1.0, 0.05, 0.1, 0.3, 0.5, 0.9, 0.95, rng.uniform()]))
        assert NO.quantile_linear(v, q) == float(np.quantile(v, q))


def test_mwm_tie_structure_documented():
    """Crossing vs nested matchings of four users have equal exact totals; the winner is decided
    by rounding -- the reason dB gains are an explicit input of the parity interface."""
    g = np.array([-130.0, -121.3, -104.2, -99.9])
    S = np.abs(g[:, None] - g[None, :])
    np.fill_diagonal(S, -np.inf)
    feas = 1 - np.eye(4)
    a = S[0, 2] + S[1, 3]
    b = S[0, 3] + S[1, 2]
    assert abs(a - b) < 1e-12
    pairs = NO.mwm_primary(S, feas, 1.0, True)
    assert len(pairs) == 2
