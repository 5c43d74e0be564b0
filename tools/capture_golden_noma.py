#!/usr/bin/env python3
"""Capture golden vectors for the NOMA grouping stage from the REFERENCE.

Runs ONLY in the CPU build container.  `Simulation-MARL-BCD/marl_train_bcd.py` cannot be
imported (it parses argv and trains at import, and needs tensorboard), so this tool parses it
with `ast` and executes, unmodified and in memory only,

  * its helper functions (`_anneal_topk`, `_build_feasible_mask_from_delta_g`,
    `_score_matrix_from_gain_and_history`, `_relax_mask_once`, `_mwm_completion`,
    `_mwm_primary`, `_adaptive_threshold_from_delta_g`, `_qos_pair_feasible`), and
  * the statements of its step loop that form the pairing stage (from the assignment of
    `offload_power_for_pairing` to the `unpaired_streak` update) plus the three statements that
    track `ep_env_best` / `last_env_global` after `env.step`,

against a namespace this tool prepares (a `config` object, recorded inputs, the episode-scoped
state variables).  Nothing of the reference's text is written anywhere: the fixtures
`tests/golden/noma_*.npz` hold inputs and outputs only.

    python tools/capture_golden_noma.py
"""
from __future__ import annotations

import ast
import math
import os
import random
import sys
import types
import typing

import numpy as np

REF = "/root/reference/Simulation-MARL-BCD/marl_train_bcd.py"
OUT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
if not os.path.isfile(REF):
    sys.exit("capture_golden_noma: reference not present (this tool only runs in the build container)")

HELPERS = ("_anneal_topk", "_build_feasible_mask_from_delta_g", "_score_matrix_from_gain_and_history",
           "_relax_mask_once", "_mwm_completion", "_mwm_primary", "_adaptive_threshold_from_delta_g",
           "_qos_pair_feasible")

TREE = ast.parse(open(REF, encoding="utf-8").read(), filename=REF)


def _compile(nodes):
    mod = ast.Module(body=list(nodes), type_ignores=[])
    return compile(mod, REF, "exec")


def base_namespace(config):
    ns = dict(np=np, math=math, random=random, config=config, Optional=typing.Optional, Set=typing.Set,
              Tuple=typing.Tuple, List=typing.List)
    defs = [n for n in TREE.body if isinstance(n, ast.FunctionDef) and n.name in HELPERS]
    assert sorted(d.name for d in defs) == sorted(HELPERS), "helper set changed in the reference"
    exec(_compile(defs), ns)
    return ns


def _assigned_names(stmt):
    out = set()
    for n in ast.walk(stmt):
        if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Store):
            out.add(n.id)
    return out


def find_step_blocks():
    """-> (pairing statements, reward-tracking statements) of the training step loop."""
    loops = [n for n in ast.walk(TREE) if isinstance(n, ast.For) and isinstance(n.target, ast.Name)
             and n.target.id == "i_step"]
    loop = next(l for l in loops if any("offload_power_for_pairing" in _assigned_names(s) and
                                        isinstance(s, ast.Assign) for s in l.body)
                and any("unstick_used_flag" in _assigned_names(s) for s in l.body))
    body = loop.body
    first = next(k for k, s in enumerate(body) if isinstance(s, ast.Assign)
                 and "offload_power_for_pairing" in _assigned_names(s))
    last = next(k for k, s in enumerate(body) if isinstance(s, ast.For) and "unpaired_streak" in ast.dump(s)
                and isinstance(s.target, ast.Name) and s.target.id == "u")
    assert first < last
    pairing = body[first:last + 1]
    # ep_env_best / last_env_global bookkeeping: `if last_env_global is None: ... else: ...` and the
    # assignment that follows it
    k = next(k for k, s in enumerate(body) if isinstance(s, ast.If) and "ep_env_best" in _assigned_names(s))
    track = body[k:k + 2]
    assert "last_env_global" in _assigned_names(track[1])
    return _compile(pairing), _compile(track)


PAIRING, TRACK = find_step_blocks()


def make_config(N, **over):
    """Class defaults of the driver's Config (attributes the pairing stage reads)."""
    c = types.SimpleNamespace(
        n_veh=N, use_mwm_completion=True, mwm_allow_singles=True, use_mwm_primary=True, mwm_accept_quantile=0.10,
        mwm_backoff_rounds=5, mwm_accept_q_step=0.05, min_pair_target=max(1, N // 4),
        pairing_threshold_quantile=0.5, mask_enable=True, mask_topk_start=N - 1, mask_topk_end=max(4, N // 2),
        mask_tau_q_start=0.2, mask_tau_q_end=0.4, mask_warmup_episodes=200, qos_enable=False,
        qos_R_min_bpsHz=0.0)
    for k, v in over.items():
        setattr(c, k, v)
    return c


YAML8 = dict(min_pair_target=3, mwm_backoff_rounds=3, qos_enable=True, qos_R_min_bpsHz=0.15, mask_topk_start=7,
             mask_topk_end=7, mask_tau_q_start=0.10, mask_tau_q_end=0.25, pairing_threshold_quantile=0.25,
             score_w_delta_db=1.0, score_w_history=0.3, stickiness_tau_db=1.0, pair_hist_decay=0.97,
             pair_unstick_prob=0.02, abs_gain_min_db=float("-inf"), relax_q_step=0.02, relax_topk_step=1,
             relax_tau_factor_per_round=0.95, freeze_group_in_episode=True, freeze_recalc_every=0,
             freeze_unstick_prob=0.0, freeze_reward_drop_ratio=0.05)


def draw_gains(rng, N, kind):
    """float32-representable linear gains (the build's gains are fp32)."""
    if kind == "wide":          # straddles the 1e-12 clamp of the score matrix
        g = 10.0 ** rng.uniform(-13.3, -10.0, N)
    elif kind == "strong":      # all above 1e-12: no clamp ties
        g = 10.0 ** rng.uniform(-11.8, -9.5, N)
    else:                       # "weak": most users clamp to -120 dB
        g = 10.0 ** rng.uniform(-13.5, -11.7, N)
    return g.astype(np.float32).astype(np.float64)


def partner_of(groups, N):
    p = np.full((N,), -2, dtype=np.int32)
    for g in groups:
        if len(g) == 1:
            p[g[0]] = -1
        elif len(g) == 2:
            p[g[0]] = g[1]
            p[g[1]] = g[0] + (1 << 16)
    return p


def capture_helpers(tag, N, seed, n_cases=96):
    rng = np.random.default_rng(seed)
    ns = base_namespace(make_config(N))
    rec = {k: [] for k in ("gain", "gdb15", "gdb12", "q", "K", "tau", "mask", "hist", "qos", "S", "accept_q",
                           "mwm_partner", "mwm_npairs", "comp_partner", "comp_npairs", "min_pairs", "relax_tau",
                           "relax_topk", "relax_mask", "row_tie", "topk_tie", "S_relaxed", "p01", "qos_ok",
                           "mwm_nosingles_partner", "mwm_nosingles_npairs", "abs_min", "S_absmin")}
    for c in range(n_cases):
        g = draw_gains(rng, N, ("wide", "strong", "weak")[c % 3])
        q = float(rng.uniform(0.05, 0.6))
        K = int(rng.integers(1, N))
        tau = ns["_adaptive_threshold_from_delta_g"](g, q)
        mask = ns["_build_feasible_mask_from_delta_g"](g, tau, K)
        gdb15 = 10.0 * np.log10(np.maximum(g, 1e-15))
        gdb12 = 10.0 * np.log10(np.maximum(g, 1e-12))
        gap15 = np.abs(gdb15[:, None] - gdb15[None, :])
        topk_tie = any(len(set(gap15[i, np.flatnonzero((gap15[i] >= tau) & (np.arange(N) != i))])) <
                       np.count_nonzero((gap15[i] >= tau) & (np.arange(N) != i)) for i in range(N))
        hist = (rng.integers(0, 4, (N, N)) * rng.uniform(0.5, 1.0)).astype(np.float32)
        hist = np.triu(hist, 1); hist = hist + hist.T
        if c % 4 == 0:
            hist[:] = 0
        p01 = rng.uniform(0, 1, N).astype(np.float32)
        noise, Pmax, Rmin = 10 ** (-174 / 10) / 1000 * 5e6, 2.0, 0.15
        qos = np.zeros((N, N), dtype=np.uint8)
        for i in range(N):
            for j in range(N):
                if i != j:
                    qos[i, j] = 1 if ns["_qos_pair_feasible"](i, j, g, p01.astype(float), noise_power=noise,
                                                              P_max=Pmax, R_min=Rmin) else 0
        use_qos = c % 2 == 0
        S = ns["_score_matrix_from_gain_and_history"](gain_linear=g, feasible_mask=mask, hist_affinity=hist,
                                                      w_delta_db=1.0, w_hist=0.3, abs_gain_min_db=-math.inf,
                                                      qos_soft_mask=qos if use_qos else None, qos_soft_penalty=6.0)
        accept_q = float(rng.choice([0.05, 0.10, 0.2, 0.5, 1.0]))
        pairs = ns["_mwm_primary"](S=S, feasible=mask, accept_quantile=accept_q, allow_singles=True)
        pairs_ns = ns["_mwm_primary"](S=S, feasible=mask, accept_quantile=accept_q, allow_singles=False)
        abs_min = float(np.quantile(gdb12, min(1.0, rng.uniform(0.2, 1.1)) if c % 3 else 0.5)) if c % 7 else 0.0
        S_abs = ns["_score_matrix_from_gain_and_history"](gain_linear=g, feasible_mask=mask, hist_affinity=hist,
                                                          w_delta_db=0.7, w_hist=0.45, abs_gain_min_db=abs_min,
                                                          qos_soft_mask=qos if use_qos else None, qos_soft_penalty=2.5)
        min_pairs = int(rng.integers(1, N // 2 + 1))
        comp = ns["_mwm_completion"](S, mask, [tuple(p) for p in pairs], min_pairs)
        rtau = float(rng.uniform(2.0, 12.0))
        rtop = int(rng.integers(1, N))
        rel = ns["_relax_mask_once"](mask.astype(np.uint8), g, rtau, rtop)
        gap12 = np.abs(gdb12[:, None] - gdb12[None, :])
        row_tie = any(len(set(gap12[i])) < N for i in range(N))
        S2 = ns["_score_matrix_from_gain_and_history"](gain_linear=g, feasible_mask=rel, hist_affinity=hist,
                                                       w_delta_db=1.0, w_hist=0.3, abs_gain_min_db=-math.inf,
                                                       qos_soft_mask=qos if use_qos else None, qos_soft_penalty=6.0)
        grp = lambda P: [list(p) for p in P] + [[k] for k in range(N) if k not in {u for p in P for u in p}]
        for k, v in dict(gain=g, gdb15=gdb15, gdb12=gdb12, q=q, K=K, tau=tau, mask=mask, hist=hist,
                         qos=qos if use_qos else np.full((N, N), 255, np.uint8), S=S, accept_q=accept_q,
                         mwm_partner=partner_of(grp(pairs), N), mwm_npairs=len(pairs),
                         comp_partner=partner_of(grp(comp), N), comp_npairs=len(comp), min_pairs=min_pairs,
                         relax_tau=rtau, relax_topk=rtop, relax_mask=rel, row_tie=row_tie, topk_tie=topk_tie,
                         S_relaxed=S2, p01=p01, qos_ok=qos, mwm_nosingles_partner=partner_of(grp(pairs_ns), N),
                         mwm_nosingles_npairs=len(pairs_ns), abs_min=abs_min, S_absmin=S_abs).items():
            rec[k].append(v)
    np.savez_compressed(os.path.join(OUT_DIR, "noma_helpers_%s.npz" % tag), N=N, noise_power=noise, P_max=Pmax,
                        R_min=Rmin, **{k: np.asarray(v) for k, v in rec.items()})
    print("noma_helpers_%s: %d cases, %d with relax-row ties" % (tag, n_cases, int(np.sum(rec["row_tie"]))))


def capture_episodes(tag, N, seed, cfg_over, n_ep=24, n_steps=8, refresh_every=3, gain_kinds=("wide", "strong", "weak"),
                     first_episode=0):
    """Drive the reference's pairing statements over short episodes; the mask is rebuilt with the
    reference's own helpers on "refresh" steps exactly as TRAIN:1319-1343 does."""
    rng = np.random.default_rng(seed)
    cfg = make_config(N, **cfg_over)
    ns0 = base_namespace(cfg)
    keys = ("gain", "gdb15", "gdb12", "policy", "mask", "has_mask", "q_now", "K_now", "tau_now", "global_reward",
            "u_unstick", "partner", "n_groups", "n_pairs", "rounds", "recomputed", "hist", "streak", "unstick_used",
            "row_tie", "S0")
    rec = {k: [] for k in keys}
    for ep in range(n_ep):
        i_episode = first_episode + ep * 37
        st = dict(pair_affinity_hist=np.zeros((N, N), dtype=np.float32), prev_pairs=set(),
                  unpaired_streak=np.zeros((N,), dtype=np.int32),
                  freeze_group_in_episode=bool(getattr(cfg, "freeze_group_in_episode", True)),
                  freeze_recalc_every=int(getattr(cfg, "freeze_recalc_every", 0)),
                  freeze_unstick_prob=float(getattr(cfg, "freeze_unstick_prob", 0.0)),
                  freeze_reward_drop_ratio=float(getattr(cfg, "freeze_reward_drop_ratio", 0.05)),
                  episode_groups=None, last_env_global=None, ep_env_best=-1e18, unstick_used_flag=False,
                  last_mask_mat=None, last_tau_now=None, last_K_now=None, last_q_now=None,
                  ep_tau_sum=0.0, ep_pair_calls=0, ep_pairs=0, ep_singles=0)
        ns = dict(ns0)
        ns.update(st)
        ns["env"] = types.SimpleNamespace(noise_power=float(cfg_over.get("_noise", 10 ** (-174 / 10) / 1000 * 1e6)),
                                          P_max=float(cfg_over.get("_pmax", 1.0)))
        ns["i_episode"] = i_episode
        step_rec = {k: [] for k in keys}
        g = None
        for i_step in range(n_steps):
            ns["i_step"] = i_step
            mask_mat = None
            q_now = K_now = tau_now = np.nan
            if i_step % refresh_every == 0:
                g = draw_gains(rng, N, gain_kinds[(ep + i_step) % len(gain_kinds)])
            if cfg.mask_enable and (i_step % refresh_every == 0 or ns["last_mask_mat"] is None):
                prog = min(1.0, i_episode / max(1, cfg.mask_warmup_episodes))
                K_now = ns["_anneal_topk"](i_episode, N, cfg.mask_topk_start, cfg.mask_topk_end,
                                           cfg.mask_warmup_episodes)
                q_now = float(cfg.mask_tau_q_start + (cfg.mask_tau_q_end - cfg.mask_tau_q_start) * prog)
                tau_now = ns["_adaptive_threshold_from_delta_g"](g, q_now)
                mask_mat = ns["_build_feasible_mask_from_delta_g"](g, tau_now, K_now)
                ns.update(last_mask_mat=mask_mat, last_tau_now=tau_now, last_K_now=K_now, last_q_now=q_now)
            policy = rng.uniform(-1.2, 1.2, (N, 2)).astype(np.float32)
            ns["marl_power_actions"] = [policy[i] for i in range(N)]
            ns["current_channel_gains"] = g
            ns["mask_mat"] = mask_mat
            draws = []
            orig_rand = np.random.rand
            np.random.rand = lambda *a: (draws.append(orig_rand(*a)) or draws[-1])
            was_groups = ns["episode_groups"]
            try:
                exec(PAIRING, ns)
            finally:
                np.random.rand = orig_rand
            groups = ns["noma_groups"]
            recomputed = not (ns["freeze_group_in_episode"] and was_groups is not None and not ns["need_repair"])
            gdb12 = 10.0 * np.log10(np.maximum(g, 1e-12))
            gap12 = np.abs(gdb12[:, None] - gdb12[None, :])
            reward = float(np.float32(-rng.uniform(0.5, 6.0) if rng.uniform() < 0.9 else rng.uniform(0.0, 1.0)))
            ns["_raw_global"] = reward
            exec(TRACK, ns)
            for k, v in dict(gain=g, gdb15=10.0 * np.log10(np.maximum(g, 1e-15)), gdb12=gdb12, policy=policy,
                             mask=mask_mat if mask_mat is not None else np.zeros((N, N), np.float32),
                             has_mask=mask_mat is not None, q_now=q_now, K_now=K_now, tau_now=tau_now,
                             global_reward=reward, u_unstick=draws[0] if draws else np.nan,
                             partner=partner_of(groups, N), n_groups=len(groups),
                             n_pairs=sum(1 for x in groups if len(x) == 2), rounds=ns["round_id"],
                             recomputed=recomputed, hist=ns["pair_affinity_hist"].copy(),
                             streak=ns["unpaired_streak"].copy(), unstick_used=ns["unstick_used_flag"],
                             row_tie=any(len(set(gap12[i])) < N for i in range(N)), S0=ns["S0"].copy()).items():
                step_rec[k].append(v)
        for k in keys:
            rec[k].append(np.asarray(step_rec[k]))
    meta = {k: v for k, v in vars(cfg).items() if isinstance(v, (int, float, bool))}
    np.savez_compressed(os.path.join(OUT_DIR, "noma_episodes_%s.npz" % tag), N=N, refresh_every=refresh_every,
                        first_episode=first_episode, noise_power=ns["env"].noise_power, P_max=ns["env"].P_max,
                        cfg_keys=np.asarray(sorted(meta)), cfg_vals=np.asarray([float(meta[k]) for k in sorted(meta)]),
                        **{k: np.asarray(v) for k, v in rec.items()})
    r = np.asarray(rec["rounds"])
    print("noma_episodes_%s: %d episodes x %d steps, recomputed %d, back-off rounds used in %d steps, tie steps %d"
          % (tag, n_ep, n_steps, int(np.sum(rec["recomputed"])), int(np.sum(r > 0)), int(np.sum(rec["row_tie"]))))


def main():
    os.makedirs(OUT_DIR, exist_ok=True)
    random.seed(1234)
    np.random.seed(1234)
    capture_helpers("4", 4, 11, 48)
    capture_helpers("8", 8, 12, 96)
    capture_helpers("16", 16, 13, 48)
    capture_episodes("default_8", 8, 21, {})
    capture_episodes("yaml_8", 8, 22, dict(YAML8, _noise=10 ** (-174 / 10) / 1000 * 5e6, _pmax=2.0))
    capture_episodes("tight_8", 8, 23, dict(min_pair_target=4, mask_topk_start=3, mask_topk_end=2, mask_tau_q_start=0.6,
                                            mask_tau_q_end=0.8, mwm_backoff_rounds=4, freeze_recalc_every=3,
                                            freeze_unstick_prob=0.3, mask_warmup_episodes=300))
    capture_episodes("nomask_8", 8, 24, dict(mask_enable=False, min_pair_target=3, freeze_group_in_episode=False),
                     n_ep=12)
    capture_episodes("default_16", 16, 25, dict(mwm_accept_quantile=0.2), n_ep=12, n_steps=6)
    capture_episodes("default_4", 4, 26, dict(min_pair_target=2), n_ep=12, n_steps=6)


if __name__ == "__main__":
    main()
