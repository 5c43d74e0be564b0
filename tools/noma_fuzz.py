#!/usr/bin/env python3
"""Extended randomised check of the pairing stage beyond 8 vehicles against the CPU oracle (the pytest fuzz uses 16 envs x 12
seeds; this runs more envs, more seeds, only N in 9..16): random pairing configurations x random gains, device == oracle on
every step, history / streak at the end.  Usage: noma_fuzz.py [n_seeds [E]]"""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import noma_oracle as NO                     # checker (this is a test tool, not product code)
from tests.test_noma_hip import StubEnv, cfg_from, random_gains, T
from ris_vec_marl_amd import NomaGrouper

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 24
E = int(sys.argv[2]) if len(sys.argv) > 2 else 48
t0 = time.time()
checked = dense = 0
for seed in range(n_seeds):
    rng = np.random.default_rng(5000 + seed)
    N = int(rng.choice([9, 11, 12, 13, 14, 15, 16, 16]))
    prm = NO.NomaParams(
        min_pair_target=int(rng.integers(1, N // 2 + 2)), mwm_allow_singles=bool(rng.integers(0, 2)),
        mwm_accept_quantile=float(rng.choice([0.05, 0.1, 0.2, 0.3, 0.5])),
        mwm_backoff_rounds=int(rng.integers(0, 5)), mwm_accept_q_step=float(rng.choice([0.05, 0.02])),
        completion_min_quantile=float(rng.choice([0.3, 0.0, 0.8])), score_w_delta_db=float(rng.choice([1.0, 0.5])),
        score_w_history=float(rng.choice([0.3, 0.0, 1.5])), abs_gain_min_db=float(rng.choice([-np.inf, -118.0, -105.0])),
        qos_enable=bool(rng.integers(0, 2)), qos_R_min_bpsHz=float(rng.choice([0.15, 1.0, 0.0])),
        qos_soft_penalty_dbscore=float(rng.choice([6.0, 1.0])), relax_topk_step=int(rng.integers(1, 3)),
        relax_tau_factor_per_round=float(rng.choice([0.95, 0.5])), tau_back_floor_db=float(rng.choice([3.0, 0.5])),
        pair_hist_decay=float(rng.choice([0.97, 0.5])), mask_enable=bool(rng.integers(0, 2)),
        mask_topk_start=N - 1, mask_topk_end=max(1, N // 2), mask_tau_q_start=0.2, mask_tau_q_end=0.6,
        mask_warmup_episodes=100, pairing_threshold_quantile=float(rng.choice([0.5, 0.25])),
        freeze_group_in_episode=bool(rng.integers(0, 4) > 0), freeze_recalc_every=int(rng.choice([0, 2])),
        freeze_unstick_prob=float(rng.choice([0.0, 0.4])), freeze_reward_drop_ratio=float(rng.choice([0.05, -5.0])),
        noise_power=10 ** (-174 / 10) / 1000 * 5e6, P_max=2.0)
    env = StubEnv(E, N, prm.noise_power, prm.P_max)
    grouper = NomaGrouper(env, cfg_from(prm, N))
    i_episode = int(rng.integers(0, 150))
    grouper.begin_episode(i_episode)
    eps = [NO.NomaEpisode(N) for _ in range(E)]
    prev = None
    for t in range(5):
        if t % 3 == 0:
            g = random_gains(rng, E, N)
            if rng.random() < 0.5:                          # a few envs with every gain at the floor: complete pairing graphs
                g[: max(1, E // 8)] = np.float32(1e-13)
        gd = T(g)
        gdb15 = (10.0 * torch.log10(torch.clamp(gd.double(), min=1e-15))).cpu().numpy()
        gdb12 = (10.0 * torch.log10(torch.clamp(gd.double(), min=1e-12))).cpu().numpy()
        refreshed = prm.mask_enable and t % 3 == 0
        if refreshed:
            grouper.refresh_mask(gain=gd)
        p01 = rng.uniform(0, 1, (E, N)).astype(np.float32)
        u = rng.uniform(0, 1, E).astype(np.float32)
        partner, ng = grouper.group(T(p01), t, gain=gd, prev_global=prev, u_unstick=T(u))
        partner, ng = partner.cpu().numpy(), ng.cpu().numpy()
        info = grouper.info.cpu().numpy()
        dense += int(((info[:, 0] == 1) & (info[:, 3] >= 15)).sum())
        reward = (-rng.uniform(0.5, 6.0, E)).astype(np.float32)
        for e in range(E):
            mask_o = NO.rebuild_mask(eps[e], gdb15[e], prm, i_episode) if refreshed else None
            groups, _ = NO.group_step(eps[e], g[e].astype(np.float64), p01[e].astype(np.float64), mask_o, prm,
                                      i_episode, t, u_unstick=float(u[e]), gdb15=gdb15[e], gdb12=gdb12[e])
            po, ngo = NO.partner_of_groups(groups, N)
            assert np.array_equal(partner[e], po), (seed, N, t, e, partner[e], po)
            assert ng[e] == ngo
            eps[e].observe_reward(float(reward[e]))
            checked += 1
        prev = T(reward)
    assert np.array_equal(grouper.pair_affinity_hist.cpu().numpy(), np.stack([x.hist for x in eps]))
    assert np.array_equal(grouper.unpaired_streak.cpu().numpy(), np.stack([x.streak for x in eps]))
    print(json.dumps(dict(seed=seed, N=N, singles=prm.mwm_allow_singles, q=prm.mwm_accept_quantile, ok=True,
                          elapsed_s=round(time.time() - t0, 1))), flush=True)
print(json.dumps(dict(noma_fuzz="ok", env_steps_checked=checked, solves_with_15_or_more_matchable_users=dense)))
