"""CPU ORACLE for the NOMA grouping stage -- TEST INFRASTRUCTURE ONLY.

float64 NumPy / plain-Python restatement of the pairing logic the reference driver
`Simulation-MARL-BCD/marl_train_bcd.py` (TRAIN below) runs immediately before every
`env.step()`: the |delta g_dB| feasibility mask (TRAIN:128-156, 842-855), the score matrix
(TRAIN:164-194), the quantile-gated max-weight matching (TRAIN:326-398), its greedy completion
(TRAIN:276-324), the mask relaxation used by the back-off rounds (TRAIN:260-275), the pair
QoS check (TRAIN:858-880) and the per-step control logic around them, including the
freeze-in-episode rule with its three safeties (TRAIN:1401-1562, 1618-1623).

    Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may
    import it.  The product package must not (and does not).

Parity status: PINNED.  `tools/capture_golden_noma.py` executes the reference's own helper
functions and the reference's own step-loop statements (extracted from the script by AST at
capture time; TRAIN cannot be imported, it trains at import) on recorded inputs and stores
inputs + outputs in `tests/golden/noma_*.npz`; `tests/test_noma_oracle_golden.py` checks every
function here against them (exact for masks / pairs / counters, bit-exact for float64 scores and
thresholds).

Two facts about the reference that shape the parity interface:

* Exact ties decided by rounding.  With scores S_ij = |g_i - g_j| (dB), a "crossing" and a
  "nested" matching of the same four users have mathematically EQUAL totals; which one the
  reference's matcher returns is decided by the last-bit rounding of `10*log10(g)`.  The
  restatement therefore accepts the dB gains as an explicit input (`gdb15`, `gdb12`: the two
  clamps the reference uses, 1e-15 at TRAIN:136-137/846-847 and 1e-12 at TRAIN:176/269), the
  same way random draws are explicit inputs elsewhere, and performs every later operation in
  the reference's association order.
* `np.argsort` is not stable on AVX-512 hosts (NumPy >= 1.25 dispatches to a SIMD network),
  so the order of EQUAL keys at TRAIN:151 and TRAIN:271 depends on the CPU the reference
  happens to run on.  Equal keys are common at TRAIN:271 (every gain below 1e-12 clamps to
  exactly -120 dB).  `stable=True` (what the HIP kernels implement: equal keys in index order,
  NumPy's portable behaviour for rows of <= 16 elements) is the defined behaviour of the
  build; `stable=False` calls `np.argsort` as the reference does, for the golden comparison on
  the capture host.

Single-env functions (N users); the tests loop over envs.
"""
from __future__ import annotations

import dataclasses
import math
from typing import List, Optional, Sequence, Tuple

import numpy as np

NEG_INF = -math.inf


# --------------------------------------------------------------------------
# parameters (driver `Config` + YAML keys, TRAIN:435-441, 489-503, 639-660, 716-741, 750-753)
# --------------------------------------------------------------------------
@dataclasses.dataclass
class NomaParams:
    min_pair_target: int = 2                 # TRAIN:489  max(1, n_veh // 4); YAML 3
    use_mwm_primary: bool = True             # TRAIN:438  (only this path is in scope)
    mwm_allow_singles: bool = True           # TRAIN:436
    mwm_accept_quantile: float = 0.10        # TRAIN:439
    mwm_backoff_rounds: int = 5              # TRAIN:440  (YAML 3)
    mwm_accept_q_step: float = 0.05          # TRAIN:441
    completion_min_quantile: float = 0.30    # TRAIN:282, 300
    score_w_delta_db: float = 1.0            # TRAIN:716
    score_w_history: float = 0.3             # TRAIN:717
    abs_gain_min_db: float = NEG_INF         # TRAIN:723
    qos_enable: bool = False                 # TRAIN:750
    qos_R_min_bpsHz: float = 0.0             # TRAIN:751
    qos_soft_penalty_dbscore: float = 6.0    # TRAIN:172, 1450
    relax_q_step: float = 0.02               # TRAIN:727
    relax_topk_step: int = 1                 # TRAIN:728
    relax_tau_factor_per_round: float = 0.95  # TRAIN:729
    tau_back_floor_db: float = 3.0           # TRAIN:1499
    pair_hist_decay: float = 0.97            # TRAIN:719
    mask_enable: bool = True                 # TRAIN:498
    mask_topk_start: int = 7                 # TRAIN:499  n_veh - 1
    mask_topk_end: int = 4                   # TRAIN:500  max(4, n_veh // 2)
    mask_tau_q_start: float = 0.2            # TRAIN:501
    mask_tau_q_end: float = 0.4              # TRAIN:502
    mask_warmup_episodes: int = 200          # TRAIN:503
    pairing_threshold_quantile: float = 0.5  # TRAIN:492
    freeze_group_in_episode: bool = True     # TRAIN:738
    freeze_recalc_every: int = 0             # TRAIN:739
    freeze_unstick_prob: float = 0.0         # TRAIN:740
    freeze_reward_drop_ratio: float = 0.05   # TRAIN:741
    noise_power: float = 10 ** (-174 / 10) / 1000 * 1e6   # env.noise_power (ENV:72-76)
    P_max: float = 1.0                       # env.P_max (ENV:125)

    @staticmethod
    def yaml_effective(n_veh: int = 8) -> "NomaParams":
        """Values in force with the shipped config.yaml (its lines 39-44, 53, 62, 116-120, 144-148
        and the `reward:` block 78-83 read at TRAIN:575, 639-644)."""
        return NomaParams(min_pair_target=3, mwm_accept_quantile=0.10, mwm_backoff_rounds=3,
                          mwm_accept_q_step=0.05, qos_enable=True, qos_R_min_bpsHz=0.15,
                          mask_topk_start=7, mask_topk_end=7, mask_tau_q_start=0.10, mask_tau_q_end=0.25,
                          mask_warmup_episodes=200, pairing_threshold_quantile=0.25,
                          noise_power=10 ** (-174 / 10) / 1000 * 5e6, P_max=2.0)


# --------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------
def gain_db(gain: np.ndarray, eps: float) -> np.ndarray:
    """10*log10(max(g, eps)) -- eps = 1e-15 at TRAIN:136-137, 846-847; 1e-12 at TRAIN:176, 212, 269."""
    return 10.0 * np.log10(np.maximum(np.asarray(gain, dtype=np.float64), eps))


def quantile_linear(values: np.ndarray, q: float) -> float:
    """np.quantile(values, q) (method 'linear') restated operation by operation, because the
    thresholds it yields are compared with `>=` against the very values they were interpolated
    from: virtual index (n-1)*q, neighbours floor / floor+1 (both the last element once the
    index reaches n-1), and the two-sided lerp NumPy uses (a + d*t below
    t = 0.5, b - d*(1-t) from 0.5 up)."""
    v = np.sort(np.asarray(values, dtype=np.float64).ravel())
    n = v.size
    q = float(q)
    vi = (n - 1) * q
    lo = math.floor(vi)
    hi = lo + 1
    if vi >= n - 1:
        lo = hi = n - 1
    if vi < 0:
        lo = hi = 0
    a, b = float(v[lo]), float(v[hi])
    t = vi - (-1.0 if vi >= n - 1 else (0.0 if vi < 0 else float(lo)))
    d = b - a
    r = a + d * t
    if t >= 0.5:
        r = b - d * (1.0 - t)
    return float(r)


def order_desc(keys: np.ndarray, stable: bool) -> np.ndarray:
    """argsort(-keys): indices by decreasing key.  stable=True: equal keys in index order."""
    return np.argsort(-np.asarray(keys), kind="stable") if stable else np.argsort(-np.asarray(keys))


def anneal_topk(i_ep: int, n_agents: int, k_start: int, k_end: int, T_ep: int) -> int:
    """TRAIN:128-132 (Python round() = half-to-even)."""
    i = max(0, min(i_ep, T_ep))
    k = round(k_end + (k_start - k_end) * (1.0 - i / max(1, T_ep)))
    return int(min(max(k, 1), n_agents - 1))


def mask_schedule(prm: NomaParams, i_episode: int, n_veh: int) -> Tuple[float, int]:
    """(q_now, K_now) of the curriculum at TRAIN:1323-1332."""
    prog = min(1.0, i_episode / max(1, prm.mask_warmup_episodes))
    K = anneal_topk(i_episode, n_veh, prm.mask_topk_start, prm.mask_topk_end, prm.mask_warmup_episodes)
    q = float(prm.mask_tau_q_start + (prm.mask_tau_q_end - prm.mask_tau_q_start) * prog)
    return q, K


def adaptive_threshold(gdb15: np.ndarray, q: float) -> float:
    """TRAIN:842-855: quantile q of |g_strong - g_weak| over the weak half x strong half."""
    g = np.asarray(gdb15, dtype=np.float64)
    n = g.size
    if n < 2:
        return 0.0
    order = np.argsort(g, kind="stable")
    weak, strong = order[: n // 2], order[n // 2:]
    if weak.size == 0 or strong.size == 0:
        return 0.0
    diffs = np.abs(g[strong][:, None] - g[weak][None, :]).ravel()
    return quantile_linear(diffs, q)


def feasible_mask(gdb15: np.ndarray, tau: float, K: int, stable: bool = True) -> np.ndarray:
    """TRAIN:134-156: keep (i,j) when |g_i - g_j| >= tau, then the K largest gaps per row, then
    AND with the transpose.  float32 0/1 matrix, zero diagonal."""
    g = np.asarray(gdb15, dtype=np.float64)
    N = g.size
    gap = np.abs(g[:, None] - g[None, :])
    m = (gap >= tau).astype(np.float32)            # "< tau -> 0"  (TRAIN:143-145)
    np.fill_diagonal(m, 0.0)
    for i in range(N):
        cand = np.flatnonzero(m[i] > 0)
        if cand.size > K:
            keep = cand[order_desc(gap[i, cand], stable)[:K]]
            row = np.zeros(N, dtype=np.float32)
            row[keep] = 1.0
            m[i] = row
    return m * m.T


def qos_pair_feasible(i: int, j: int, g: np.ndarray, p01: np.ndarray, noise_power: float, P_max: float,
                      R_min: float) -> bool:
    """TRAIN:858-880: both users of the pair reach R_min (near user = larger gain, ties -> i)."""
    pi, pj = float(p01[i]) * float(P_max), float(p01[j]) * float(P_max)
    gi, gj = float(g[i]), float(g[j])
    if gi >= gj:
        gn, gf, pn, pf = gi, gj, pi, pj
    else:
        gn, gf, pn, pf = gj, gi, pj, pi
    sinr_far = (pf * gf) / (pn * gf + float(noise_power) + 1e-12)
    r_far = np.log2(1.0 + max(0.0, sinr_far))
    sinr_near = (pn * gn) / (float(noise_power) + 1e-12)
    r_near = np.log2(1.0 + max(0.0, sinr_near))
    return bool((r_far >= float(R_min)) and (r_near >= float(R_min)))


def qos_soft_mask(g: np.ndarray, p01: np.ndarray, prm: NomaParams) -> np.ndarray:
    """TRAIN:1426-1441 (zero diagonal)."""
    N = len(g)
    out = np.zeros((N, N), dtype=np.uint8)
    for i in range(N):
        for j in range(N):
            if i != j:
                out[i, j] = 1 if qos_pair_feasible(i, j, g, p01, prm.noise_power, prm.P_max,
                                                   prm.qos_R_min_bpsHz) else 0
    return out


def score_matrix(gdb12: np.ndarray, feasible: np.ndarray, hist: np.ndarray, prm: NomaParams,
                 qos: Optional[np.ndarray]) -> np.ndarray:
    """TRAIN:164-194.  `hist` is float32 and `w_hist * hist` is a float32 product (the Python
    scalar is weak under NEP 50), widened only by the addition."""
    g = np.asarray(gdb12, dtype=np.float64)
    gap = np.abs(g[:, None] - g[None, :])
    abs_ok = (g[:, None] >= prm.abs_gain_min_db) | (g[None, :] >= prm.abs_gain_min_db)
    hist_term = (np.float32(prm.score_w_history) * np.asarray(hist, dtype=np.float32)).astype(np.float64)
    S = prm.score_w_delta_db * gap + hist_term
    if not np.any((feasible > 0) & abs_ok):
        abs_ok = np.ones_like(abs_ok)
    S = np.where((feasible > 0) & abs_ok, S, NEG_INF)
    if qos is not None:
        S = np.where((qos <= 0) & np.isfinite(S), S - float(prm.qos_soft_penalty_dbscore), S)
    np.fill_diagonal(S, NEG_INF)
    return S


def relax_mask_once(mask: np.ndarray, gdb12: np.ndarray, tau_db: float, topk: int, stable: bool = True) -> np.ndarray:
    """TRAIN:260-275: OR the mask with (gap >= tau_db, off-diagonal) and with each row's `topk`
    largest gaps (the row includes the user itself, gap 0)."""
    g = np.asarray(gdb12, dtype=np.float64)
    N = g.size
    gap = np.abs(g[:, None] - g[None, :])
    cand = gap >= tau_db
    np.fill_diagonal(cand, False)
    top = np.zeros((N, N), dtype=bool)
    if topk >= 1:
        for i in range(N):
            top[i, order_desc(gap[i], stable)[: min(topk, N - 1)]] = True
    return ((np.asarray(mask) > 0) | cand | top).astype(np.uint8)


def mwm_primary(S: np.ndarray, feasible: np.ndarray, accept_quantile: float, allow_singles: bool = True
                ) -> List[Tuple[int, int]]:
    """TRAIN:326-398: exact max-weight matching over the edges whose score reaches the
    (1 - accept_quantile) quantile of all finite feasible entries (both triangles count for the
    quantile; only i<j entries become edges).

    The reference recurses on the lowest unused user x: first "x stays single", then partners j
    in increasing order, a candidate replacing the incumbent only when STRICTLY heavier; totals
    are formed as w_xj + total(rest).  Here the same recurrence is evaluated bottom-up over
    bitmasks (every mask only needs numerically larger masks), after dropping users with no
    admissible edge -- for those the recurrence passes `total(rest)` through unchanged, so the
    result is bit-identical (only done when singles are allowed; without singles such a user
    voids its branch, which the full-width table reproduces).  Returns sorted (i, j), i < j."""
    N = S.shape[0]
    edge_ok = (np.asarray(feasible) > 0) & np.isfinite(S)
    vals = S[edge_ok]
    if vals.size == 0:
        return []
    q = float(min(max(accept_quantile, 0.0), 1.0))
    thr = quantile_linear(vals, 1.0 - q)
    W = np.where((S >= thr) & edge_ok, S, NEG_INF)
    live = [v for v in range(N)
            if any(np.isfinite(W[min(v, u), max(v, u)]) for u in range(N) if u != v)] if allow_singles else list(range(N))
    K = len(live)
    if K == 0:
        return []
    w = [[W[live[a], live[b]] if a < b else NEG_INF for b in range(K)] for a in range(K)]
    full = (1 << K) - 1
    total = [0.0] * (full + 1)

    def best_at(mask: int) -> Tuple[float, int]:
        """(value, partner) for the lowest unused user of `mask`: partner -1 = single, -2 = void."""
        x = 0
        while mask >> x & 1:
            x += 1
        best, arg = NEG_INF, -2
        if allow_singles:
            w1 = total[mask | 1 << x]
            if w1 > best:
                best, arg = w1, -1
        for j in range(x + 1, K):
            if not mask >> j & 1 and math.isfinite(w[x][j]):
                w2 = total[mask | 1 << x | 1 << j]
                if math.isfinite(w2) and w[x][j] + w2 > best:
                    best, arg = w[x][j] + w2, j
        return best, arg

    for mask in range(full - 1, -1, -1):
        best, arg = best_at(mask)
        total[mask] = 0.0 if arg == -2 else best       # TRAIN:389-390
    pairs, mask = [], 0
    while mask != full:
        _, arg = best_at(mask)
        if arg == -2:
            break                                       # void branch: (0.0, []) drops what would follow
        x = 0
        while mask >> x & 1:
            x += 1
        mask |= 1 << x
        if arg >= 0:
            mask |= 1 << arg
            pairs.append((live[x], live[arg]))
    return sorted(pairs)


def mwm_completion(S: np.ndarray, feasible: np.ndarray, pairs_now: Sequence[Tuple[int, int]], min_pairs: int,
                   completion_min_quantile: float) -> List[Tuple[int, int]]:
    """TRAIN:276-324: greedy top-up, heaviest first (ties: larger (i, j) first, it is a reversed
    tuple sort), restricted to finite feasible i<j entries at or above the quantile."""
    N = S.shape[0]
    pairs = [tuple(p) for p in pairs_now]
    finite = np.isfinite(S) & (np.asarray(feasible) > 0)
    if not finite.any():
        return pairs
    thr = quantile_linear(S[finite], completion_min_quantile)
    cand = [(float(S[i, j]), i, j) for i in range(N) for j in range(i + 1, N) if finite[i, j] and S[i, j] >= thr]
    cand.sort(reverse=True)
    busy = {u for p in pairs for u in p}
    added = 0
    for _, i, j in cand:
        if i in busy or j in busy:
            continue
        pairs.append((i, j))
        busy.update((i, j))
        added += 1
        if len(pairs) >= int(min_pairs):
            break
    return pairs


# --------------------------------------------------------------------------
# per-episode state + the per-step control logic
# --------------------------------------------------------------------------
class NomaEpisode:
    """Episode-scoped variables of TRAIN:1282-1300."""

    def __init__(self, N: int):
        self.N = N
        self.hist = np.zeros((N, N), dtype=np.float32)       # pair_affinity_hist
        self.streak = np.zeros((N,), dtype=np.int32)         # unpaired_streak
        self.groups: Optional[List[List[int]]] = None        # episode_groups
        self.last_global: Optional[float] = None             # last_env_global
        self.best_global = -1e18                             # ep_env_best
        self.unstick_used = False                            # unstick_used_flag
        self.last_q: Optional[float] = None                  # last_q_now / last_K_now / last_tau_now
        self.last_K: Optional[int] = None
        self.last_tau: Optional[float] = None

    def observe_reward(self, g: float) -> None:
        """TRAIN:1618-1623."""
        if self.last_global is None:
            self.best_global = g
        elif g > self.best_global:
            self.best_global = g
        self.last_global = g


def solve_pairs(gain: np.ndarray, gdb15: np.ndarray, gdb12: np.ndarray, p01: np.ndarray,
                mask: Optional[np.ndarray], hist: np.ndarray, prm: NomaParams, q_back: float, K_back: int,
                tau_back: float, stable: bool = True) -> Tuple[List[Tuple[int, int]], int]:
    """TRAIN:1419-1524: score, primary matching, completion, then up to `mwm_backoff_rounds`
    relax-and-retry rounds while fewer than `min_pair_target` pairs exist.  -> (pairs, rounds)."""
    N = len(gain)
    feas = (np.asarray(mask).astype(np.uint8) if (prm.mask_enable and mask is not None)
            else (np.ones((N, N), dtype=np.uint8) - np.eye(N, dtype=np.uint8)))
    qos = qos_soft_mask(gain, p01, prm) if prm.qos_enable else None
    target = max(1, prm.min_pair_target)
    accept_q = float(prm.mwm_accept_quantile)
    S = score_matrix(gdb12, feas, hist, prm, qos)
    pairs = mwm_primary(S, feas, accept_q, prm.mwm_allow_singles)
    if len(pairs) < target:
        pairs = mwm_completion(S, feas, pairs, target, prm.completion_min_quantile)
    rounds = 0
    q_back, K_back, tau_back = float(q_back), int(K_back), float(tau_back)
    while len(pairs) < target and rounds < int(prm.mwm_backoff_rounds):
        rounds += 1
        q_back = max(0.05, q_back - prm.relax_q_step)          # computed but only K / tau reach the mask
        K_back = min(N - 1, K_back + prm.relax_topk_step)
        tau_back = max(float(prm.tau_back_floor_db), tau_back * prm.relax_tau_factor_per_round)
        feas = relax_mask_once(feas, gdb12, tau_back, K_back, stable)
        S = score_matrix(gdb12, feas, hist, prm, qos)
        accept_q = max(0.05, accept_q - float(prm.mwm_accept_q_step))
        pairs = mwm_primary(S, feas, accept_q, prm.mwm_allow_singles)
        if len(pairs) < target:
            pairs = mwm_completion(S, feas, pairs, target, prm.completion_min_quantile)
    return [tuple(p) for p in pairs], rounds


def group_step(ep: NomaEpisode, gain: np.ndarray, p01: np.ndarray, mask: Optional[np.ndarray], prm: NomaParams,
               i_episode: int, i_step: int, u_unstick: Optional[float] = None, stable: bool = True,
               gdb15: Optional[np.ndarray] = None, gdb12: Optional[np.ndarray] = None
               ) -> Tuple[List[List[int]], dict]:
    """One pass of TRAIN:1401-1562 for one env.  `mask` is this step's `mask_mat` (None on steps
    where the driver did not rebuild it, TRAIN:1317-1340 -- the pairing then sees the full
    off-diagonal mask, TRAIN:1421-1424).  Returns (noma_groups, info)."""
    N = ep.N
    gain = np.asarray(gain, dtype=np.float64)
    gdb15 = gain_db(gain, 1e-15) if gdb15 is None else np.asarray(gdb15, dtype=np.float64)
    gdb12 = gain_db(gain, 1e-12) if gdb12 is None else np.asarray(gdb12, dtype=np.float64)
    ep.hist *= np.float32(prm.pair_hist_decay)                  # TRAIN:1406 (float32 in place)
    if ep.last_q is not None:                                   # TRAIN:1486-1491
        q_back, K_back, tau_back = ep.last_q, ep.last_K, ep.last_tau
    else:
        q_back = prm.pairing_threshold_quantile
        K_back = anneal_topk(i_episode, N, prm.mask_topk_start, prm.mask_topk_end, prm.mask_warmup_episodes)
        tau_back = adaptive_threshold(gdb15, q_back)
    frozen = prm.freeze_group_in_episode and ep.groups is not None
    need_repair = False
    if frozen:                                                  # TRAIN:1527-1540
        if prm.freeze_recalc_every > 0 and i_step % prm.freeze_recalc_every == 0:
            need_repair = True
        if not need_repair and ep.last_global is not None and not ep.unstick_used:
            if ep.last_global < ep.best_global * (1.0 - prm.freeze_reward_drop_ratio):
                need_repair = True
                ep.unstick_used = True
        if not need_repair and prm.freeze_unstick_prob > 0.0:
            if float(u_unstick) < prm.freeze_unstick_prob:
                need_repair = True
    rounds = 0
    if frozen and not need_repair:                              # TRAIN:1542-1547
        pairs = [(g[0], g[1]) for g in ep.groups if len(g) == 2]
        recomputed = False
    else:                                                       # TRAIN:1548-1553
        pairs, rounds = solve_pairs(gain, gdb15, gdb12, p01, mask, ep.hist, prm, q_back, K_back, tau_back, stable)
        used = {u for p in pairs for u in p}
        ep.groups = [[i, j] for (i, j) in pairs] + [[k] for k in range(N) if k not in used]
        recomputed = True
    used = {u for p in pairs for u in p}
    for (i, j) in pairs:                                        # TRAIN:1556-1561
        ep.hist[i, j] += np.float32(1.0)
        ep.hist[j, i] += np.float32(1.0)
    for u in range(N):
        ep.streak[u] = 0 if u in used else ep.streak[u] + 1
    return [list(g) for g in ep.groups], dict(recomputed=recomputed, rounds=rounds, n_pairs=len(pairs))


def rebuild_mask(ep: NomaEpisode, gdb15: np.ndarray, prm: NomaParams, i_episode: int, stable: bool = True
                 ) -> np.ndarray:
    """TRAIN:1319-1343: threshold + mask rebuild on a channel-refresh step; caches (q, K, tau)."""
    q_now, K_now = mask_schedule(prm, i_episode, ep.N)
    tau = adaptive_threshold(gdb15, q_now)
    ep.last_q, ep.last_K, ep.last_tau = q_now, K_now, tau
    return feasible_mask(gdb15, tau, K_now, stable)


def partner_of_groups(groups: Sequence[Sequence[int]], N: int) -> Tuple[np.ndarray, int]:
    """Batched group encoding of the step kernels (same convention as risvec_oracle.encode_groups)."""
    partner = np.full((N,), -2, dtype=np.int32)
    for g in groups:
        if len(g) == 1:
            partner[g[0]] = -1
        elif len(g) == 2:
            partner[g[0]] = g[1]
            partner[g[1]] = g[0] + (1 << 16)
    return partner, len(groups)
