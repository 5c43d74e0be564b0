"""oracle/policy_oracle.py against vectors captured from the reference's own PolicyNetwork
(tools/capture_golden_policy.py).  The reference computes in float32, the oracle in float64:
1e-5 absolute on outputs in [-1, 1] / [0, 1]; one-hot exact wherever the top two probabilities
are further apart than that.  CPU only."""
import glob
import os

import numpy as np
import pytest

from oracle import policy_oracle as PO

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def agent_weights(d, a):
    pre = "a%d." % a
    return {k[len(pre):]: d[k] for k in d.files if k.startswith(pre)}


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "policy_*.npz"))), ids=os.path.basename)
def test_choose_action(path):
    d = np.load(path)
    V = int(d["V"])
    for a in range(V):
        w = agent_weights(d, a)
        mu, log_std, logits = PO.forward(w, d["state"][a])
        np.testing.assert_allclose(mu, d["mu"][a], atol=2e-6)
        np.testing.assert_allclose(log_std, d["log_std"][a], atol=2e-6)
        np.testing.assert_allclose(logits, d["logits"][a], atol=2e-6)
        mask = d["mask"][a] if d["has_mask"][a] else None
        hard = bool(d["hard"][a])
        power, y, onehot = PO.choose_action(w, d["state"][a], mask, float(d["tau"][a]), d["eps"][a], d["expo"][a], hard)
        np.testing.assert_allclose(power, d["power"][a], atol=1e-5)
        soft = PO.choose_action(w, d["state"][a], mask, float(d["tau"][a]), d["eps"][a], d["expo"][a])[1]
        clear = PO.top2_gap(soft) > 1e-4
        assert clear.mean() > 0.9
        assert np.array_equal(onehot[clear], d["onehot"][a][clear])
        if hard:                                      # straight-through: one-hot up to one float32 rounding
            np.testing.assert_allclose(y[clear], d["probs"][a][clear], atol=2e-7)
            assert np.array_equal(np.round(d["probs"][a]), d["onehot"][a])
            continue
        np.testing.assert_allclose(y, d["probs"][a], atol=1e-5)
        if mask is not None:                          # blocked users get (numerically) zero probability,
            m = mask.copy(); m[m.sum(-1) == 0] = 1     # except on the all-zero row the reference opens up
            assert np.all(y[m <= 0] < 1e-30)
            assert y[0].min() > 0


def test_fixture_count():
    assert len(glob.glob(os.path.join(GOLD, "policy_*.npz"))) == 2
