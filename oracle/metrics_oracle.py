"""CPU ORACLE for the per-episode metrics sink -- TEST INFRASTRUCTURE ONLY.

float64 restatement of the bookkeeping `Simulation-MARL-BCD/marl_train_bcd.py` (TRAIN below) does
around `env.step` for ONE env: the step-wise sums of the env's `last_*` scalars (TRAIN:1626-1662),
the clipped per-user rewards (TRAIN:1714, 1769), the equivalent powers from `last_power_W`
(TRAIN:1717-1753), the best global reward of the episode (TRAIN:1613-1622), and the episode-end
scalars (TRAIN:1824, 1838-1865, 1939-1941) with `_jain_index` (TRAIN:112-119).

    Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may
    import it.  The product package must not (and does not).

Parity status: PINNED by `tests/golden/episode_metrics_8.npz` (tools/capture_golden_metrics.py steps
the reference's own `Environ` and keeps the sums with the driver's statements; the driver script
itself cannot be imported -- it trains at import and needs tensorboard -- so its `_jain_index` is
compiled from the script's syntax tree and called on the captured vectors).
"""
from __future__ import annotations

from typing import Dict

import numpy as np

# column order = RISVEC_EP_* (include/risvec.h); metric slot order = RISVEC_METRIC_*
COLUMNS = (
    "reward/global_avg", "traffic/offload_kbit_ep", "traffic/local_kbit_ep", "queue/mec_cycles",
    "queue/backlog_kbit_ep_mean", "delay/local_ep_mean", "delay/edge_queue_ep_mean", "delay/edge_compute_ep_mean",
    "delay/tx_ep_mean", "queue/mec_util_ep_mean", "cpu/local_util_ep_mean", "qos/violation_rate_ep_mean",
    "delay/episode_mean", "energy/episode_mean", "power/offload_avg", "power/local_avg", "power/total_avg",
    "reward/min_user", "reward/var_user", "reward/jain", "reward/best_global",
)


def jain_index(x: np.ndarray) -> float:
    """TRAIN:112-119."""
    x = np.asarray(x, dtype=np.float64)
    if x.size == 0:
        return 0.0
    s = x.sum()
    return float(s * s / (x.size * np.square(x).sum() + 1e-12))


class EpisodeOracle:
    def __init__(self, n_veh: int, user_clip: float = 5.0):
        self.V, self.clip = int(n_veh), float(user_clip)
        self.begin_episode()

    def begin_episode(self) -> None:
        self.sums = np.zeros(14)
        self.user = np.zeros(self.V)
        self.p_off, self.p_loc, self.p_tot = [], [], []
        self.best = None
        self.steps = 0
        self.last_q = 0.0

    def accumulate(self, metrics: np.ndarray, reward: np.ndarray, power_w=None) -> None:
        """metrics: the 14 slots (global_reward, then the 13 last_*); reward [V]; power_w [2,V]."""
        m = np.asarray(metrics, dtype=np.float64)
        g = float(m[0])
        self.best = g if self.best is None or g > self.best else self.best       # TRAIN:1613-1622
        self.sums += m[:14]                                                       # TRAIN:1626-1662
        self.last_q = float(m[3])
        self.user += np.clip(np.asarray(reward, dtype=np.float64), -self.clip, self.clip)   # TRAIN:1714, 1769
        if power_w is not None:
            pw = np.asarray(power_w, dtype=np.float64)
            self.p_tot.append(float(pw.sum()))                                    # TRAIN:1719, 1747
            self.p_off.append(float(pw[0, :].sum()))                              # TRAIN:1752
            self.p_loc.append(float(pw[1, :].sum()))                              # TRAIN:1753
        self.steps += 1

    def end_episode(self) -> Dict[str, float]:
        n = self.steps
        c = list(self.sums / n)                                                   # TRAIN:1838, 1850-1865
        c[1], c[2], c[3] = float(self.sums[1]), float(self.sums[2]), self.last_q  # TRAIN:2046-2048
        user = self.user / n                                                      # TRAIN:1824
        c += [float(np.mean(self.p_off)) if self.p_off else 0.0, float(np.mean(self.p_loc)) if self.p_loc else 0.0,
              float(np.mean(self.p_tot)) if self.p_tot else 0.0,                  # TRAIN:1841-1843
              float(np.min(user)), float(np.var(user)), jain_index(user), float(self.best)]   # TRAIN:1939-1941
        return dict(zip(COLUMNS, c))
