"""oracle/replay_oracle.py against vectors captured from the reference's own ReplayBuffer and its
marshalling statements (tools/capture_golden_replay.py).  Bit-exact.  CPU only."""
import glob
import os

import numpy as np
import pytest

from oracle import replay_oracle as RO

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "replay_buffer_*.npz"))), ids=os.path.basename)
def test_ring_buffer(path):
    d = np.load(path)
    V, cap = int(d["V"]), int(d["cap"])
    buf = RO.ReplayOracle(cap, 5, V + 2, V)
    n = len(d["state"])
    for lo in range(0, n, 37):                      # ragged batches, crossing the wrap point
        hi = min(n, lo + 37)
        buf.store_batch(d["state"][lo:hi], d["action"][lo:hi], d["reward_g"][lo:hi], d["reward_l"][lo:hi],
                        d["state_"][lo:hi], d["done"][lo:hi], d["mask"][lo:hi])
    assert buf.mem_cntr == int(d["mem_cntr"])
    for k in ("state_memory", "action_memory", "reward_global_memory", "reward_local_memory", "new_state_memory",
              "terminal_memory", "mask_memory"):
        assert np.array_equal(getattr(buf, k), d[k]), k
    assert d["batch"].max() < buf.max_mem()
    out = buf.sample(d["batch"])
    for got, k in zip(out, ("s_states", "s_actions", "s_rewards_g", "s_rewards_l", "s_states_", "s_dones", "s_masks")):
        assert np.array_equal(got, d[k]), k


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "replay_marshal_*.npz"))), ids=os.path.basename)
def test_marshalling(path):
    d = np.load(path)
    for c in range(len(d["power"])):
        env_a, p01, store = RO.marshal_actions(d["power"][c][None], d["probs"][c][None], float(d["floor"][c]))
        assert np.array_equal(env_a[0], d["action_env"][c]), c
        assert np.array_equal(p01[0], d["action_env"][c][0])
        assert np.array_equal(store[0], d["store"][c]), c


def test_fixture_count():
    assert len(glob.glob(os.path.join(GOLD, "replay_*.npz"))) == 5
