"""GPU parity tests of the replay ring buffer + marshalling: bit-exact against
the vectors captured from the reference's own buffer.py / marshalling statements and against
oracle/replay_oracle.py at BASELINE batch sizes."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from oracle import replay_oracle as RO  # noqa: E402  (checker)

GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"
ARRAYS = ("state_memory", "action_memory", "reward_global_memory", "reward_local_memory", "new_state_memory",
          "terminal_memory", "mask_memory")


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "replay_buffer_*.npz"))), ids=os.path.basename)
@pytest.mark.parametrize("chunk", [1, 37])
def test_ring_vs_golden(path, chunk):
    from ris_vec_marl_amd import VecReplayBuffer
    d = np.load(path)
    V, cap = int(d["V"]), int(d["cap"])
    buf = VecReplayBuffer(cap, 5, V + 2, V, device=DEV)
    n = len(d["state"])
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        if chunk == 1:          # the reference's own signature, one transition at a time
            buf.store_transition(d["state"][lo], d["action"][lo], float(d["reward_g"][lo]), d["reward_l"][lo],
                                 d["state_"][lo], bool(d["done"][lo]), d["mask"][lo])
        else:
            buf.store_batch(T(d["state"][lo:hi]), T(d["action"][lo:hi]), T(d["reward_g"][lo:hi]), T(d["reward_l"][lo:hi]),
                            T(d["state_"][lo:hi]), T(d["done"][lo:hi]), T(d["mask"][lo:hi].reshape(-1, V, V)))
    assert buf.mem_cntr == int(d["mem_cntr"])
    for k in ARRAYS:
        assert np.array_equal(getattr(buf, k).cpu().numpy(), d[k]), k
    out = buf.sample_buffer(len(d["batch"]), idx=T(d["batch"].astype(np.int64)))
    for got, k in zip(out, ("s_states", "s_actions", "s_rewards_g", "s_rewards_l", "s_states_", "s_dones", "s_masks")):
        assert np.array_equal(got.cpu().numpy(), d[k]), k


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "replay_marshal_*.npz"))), ids=os.path.basename)
def test_marshal_vs_golden(path):
    from ris_vec_marl_amd import marshal_actions
    d = np.load(path)
    for fl in np.unique(d["floor"]):
        sel = d["floor"] == fl
        env_a, p01, store = marshal_actions(T(d["power"][sel]), T(d["probs"][sel]), float(fl))
        # the reference holds float32-valued numbers in a float64 array; the floor itself is a double
        assert np.array_equal(env_a.cpu().numpy(), d["action_env"][sel].astype(np.float32))
        assert np.array_equal(p01.cpu().numpy(), d["action_env"][sel][:, 0, :].astype(np.float32))
        assert np.array_equal(store.cpu().numpy(), d["store"][sel])


def test_full_size_from_env_outputs():
    """E = 32 768 transitions per step taken in place from a stepped VecEnviron (obs, reward,
    metrics[:,0], NOMA mask), 5 steps into a ring that wraps; device Philox sampling; all against
    the oracle fed the same arrays."""
    from ris_vec_marl_amd import NomaGrouper, VecEnviron, VecReplayBuffer, marshal_actions, reference_lanes
    E, V, M = 32768, 8, 64
    L = reference_lanes()
    env = VecEnviron(L["down_lanes"], L["up_lanes"], L["left_lanes"], L["right_lanes"], 400, 400, V, M, 3,
                     n_envs=E, device=DEV, seed=9)
    env.make_new_game(); env.renew_positions(); env.compute_parms(); env.Random_phase(); env.update_channel_gains()
    grouper = NomaGrouper(env)
    grouper.begin_episode(0)
    cap = 3 * E + 1000
    buf = VecReplayBuffer(cap, 5, V + 2, V, device=DEV, seed=123)
    orc = RO.ReplayOracle(cap, 5, V + 2, V)
    rng = np.random.default_rng(5)
    obs_old = env.tensors["obs"].clone()
    for t in range(5):
        power = T(rng.uniform(-1.1, 1.1, (E, V, 2)).astype(np.float32))
        probs = T(rng.dirichlet(np.ones(V), (E, V)).astype(np.float32))
        action_env, p01, store = marshal_actions(power, probs, env.cpu_share_floor)
        a_o, p_o, s_o = RO.marshal_actions(power.cpu().numpy(), probs.cpu().numpy(), env.cpu_share_floor)
        assert np.array_equal(action_env.cpu().numpy(), a_o.astype(np.float32))
        assert np.array_equal(store.cpu().numpy(), s_o)
        mask = grouper.refresh_mask() if t % 2 == 0 else None
        partner, ng = grouper.group(p01, t)
        env.step(action_env, partner, ng, None, fused=True)
        obs_new, reward, metrics = env.tensors["obs"], env.tensors["reward"], env.tensors["metrics"]
        done = t == 4
        buf.store_batch(obs_old, store, metrics, reward, obs_new, done, mask)
        orc.store_batch(obs_old.cpu().numpy().reshape(E, -1), s_o, metrics[:, 0].cpu().numpy(), reward.cpu().numpy(),
                        obs_new.cpu().numpy().reshape(E, -1), done,
                        None if mask is None else mask.cpu().numpy().reshape(E, -1).astype(np.float32))
        obs_old = obs_new.clone()
    assert buf.mem_cntr == orc.mem_cntr == 5 * E
    for k in ARRAYS:
        assert np.array_equal(getattr(buf, k).cpu().numpy(), getattr(orc, k)), k
    out = buf.sample_buffer(4096)
    rows = buf.last_batch.cpu().numpy()
    assert np.array_equal(rows, RO.philox_sample_indices(4096, orc.max_mem(), 1, 123))
    for got, exp in zip(out, orc.sample(rows)):
        assert np.array_equal(got.cpu().numpy(), exp)


def test_errors():
    from ris_vec_marl_amd import VecReplayBuffer
    buf = VecReplayBuffer(16, 5, 6, 4, device=DEV)
    with pytest.raises(ValueError):
        buf.sample_buffer(4)                                   # empty
    z = torch.zeros(17, 20, device=DEV)
    with pytest.raises(ValueError):                            # a batch may not overwrite itself
        buf.store_batch(z, torch.zeros(17, 24, device=DEV), torch.zeros(17, device=DEV), torch.zeros(17, 4, device=DEV), z)


@pytest.mark.parametrize("V", [6, 8, 3])
def test_bound_store_with_carry_and_odd_shapes(V):
    """bind_store(state=None): the buffer carries state_ forward itself (ping-pong copy made inside
    the store kernel).  V = 6 / 3 take the scalar (non-float4) copy path; wraps included."""
    from ris_vec_marl_amd import VecReplayBuffer
    E, cap = 1000, 3500
    rng = np.random.default_rng(V)
    buf = VecReplayBuffer(cap, 5, V + 2, V, device=DEV)
    orc = RO.ReplayOracle(cap, 5, V + 2, V)
    obs = T(rng.normal(size=(E, V, 5)).astype(np.float32))
    action = T(rng.normal(size=(E, V * (V + 2))).astype(np.float32))
    metrics = T(rng.normal(size=(E, 16)).astype(np.float32))
    reward = T(rng.normal(size=(E, V)).astype(np.float32))
    mask = T((rng.uniform(size=(E, V, V)) < 0.5).astype(np.uint8))
    store = buf.bind_store(None, action, metrics, reward, obs, mask)
    prev = obs.cpu().numpy().reshape(E, -1).copy()
    for t in range(5):
        for x in (obs, action, metrics, reward):
            x.copy_(T(rng.normal(size=tuple(x.shape)).astype(np.float32)))
        use_mask = t % 2 == 0
        store(done=t == 4, use_mask=use_mask)
        cur = obs.cpu().numpy().reshape(E, -1)
        orc.store_batch(prev, action.cpu().numpy(), metrics[:, 0].cpu().numpy(), reward.cpu().numpy(), cur, t == 4,
                        mask.cpu().numpy().reshape(E, -1).astype(np.float32) if use_mask else None)
        prev = cur.copy()
    assert buf.mem_cntr == orc.mem_cntr
    for k in ARRAYS:
        assert np.array_equal(getattr(buf, k).cpu().numpy(), getattr(orc, k)), k


@pytest.mark.parametrize("V", [5, 6, 16])
def test_marshal_vs_oracle_other_shapes(V):
    """Odd V takes the word-per-lane kernel, even V the row-per-lane one."""
    from ris_vec_marl_amd import marshal_actions
    rng = np.random.default_rng(V)
    E = 777
    power = rng.uniform(-1.2, 1.2, (E, V, 2)).astype(np.float32)
    probs = rng.dirichlet(np.ones(V), (E, V)).astype(np.float32)
    env_a, p01, store = marshal_actions(T(power), T(probs), 0.1)
    a_o, p_o, s_o = RO.marshal_actions(power, probs, 0.1)
    assert np.array_equal(env_a.cpu().numpy(), a_o.astype(np.float32))
    assert np.array_equal(p01.cpu().numpy(), p_o.astype(np.float32))
    assert np.array_equal(store.cpu().numpy(), s_o)


@pytest.mark.parametrize("V,M", [(8, 36), (6, 16), (5, 16)])
def test_rollout_without_a_marshal_launch_is_identical(V, M):
    """The three consumers of the policy outputs can take them directly -- the env through
    RISVEC_STEP_POLICY_ACTION, the grouping through risvec_noma_group_raw, the replay through
    risvec_replay_store_policy -- and must then produce exactly what the marshalled path produces
    (marl_train_bcd.py:1386-1396, 1601-1608, 1776-1784): same groups, same env state, same ring contents."""
    from ris_vec_marl_amd import NomaGrouper, VecEnviron, VecReplayBuffer, apply_yaml_config, marshal_actions, reference_lanes
    E, T = 700, 12
    L = reference_lanes()
    gen = torch.Generator(device=DEV); gen.manual_seed(3)
    power = [torch.rand(E, V, 2, device=DEV, generator=gen) * 2.4 - 1.2 for _ in range(T)]
    probs = [torch.softmax(torch.randn(E, V, V, device=DEV, generator=gen), -1) for _ in range(T)]

    def build():
        env = VecEnviron(L["down_lanes"], L["up_lanes"], L["left_lanes"], L["right_lanes"], 400, 400, V, M, 3, n_envs=E,
                         device=DEV, seed=8)
        apply_yaml_config(env, None)
        env.make_new_game(); env.renew_positions(); env.compute_parms(); env.Random_phase(); env.update_channel_gains()
        g = NomaGrouper(env)
        g.config.qos_enable = True; g.config.qos_R_min_bpsHz = 0.15; g.config.min_pair_target = max(1, V // 3)
        return env, g, VecReplayBuffer(5 * E, 5, V + 2, V, device=DEV)

    # (a) the marshalled path
    env, g, buf = build()
    g.begin_episode(0); mask = g.refresh_mask()
    state_old = env.observe().clone()
    for t in range(T):
        a_env, p01, a_store = marshal_actions(power[t], probs[t], float(env.cpu_share_floor))
        partner, ng = g.group(p01, t)
        env.step(a_env, partner, ng, None, fused=True)
        buf.store_batch(state_old, a_store, env.tensors["metrics"], env.tensors["reward"], env.tensors["obs"], t == T - 1, mask)
        state_old.copy_(env.tensors["obs"])
    # (b) no marshalling launch: bound launchers reading the policy outputs in place
    env2, g2, buf2 = build()
    g2.begin_episode(0); mask2 = g2.refresh_mask()
    pw, pr = torch.empty(E, V, 2, device=DEV), torch.empty(E, V, V, device=DEV)
    pw.copy_(power[0])
    partner2, ng2 = g2.group(None, 0, power_raw=pw)             # unbound form once: gives the state views to bind
    step = env2.bind_step(pw, partner2, ng2, None, fused=True, policy_action=True)
    group = g2.bind_group(power_raw=pw)
    store = buf2.bind_store(None, None, env2.tensors["metrics"], env2.tensors["reward"], env2.tensors["obs"], mask2,
                            policy_out=(pw, pr))
    for t in range(T):
        pw.copy_(power[t]); pr.copy_(probs[t])
        if t > 0:
            group(t)
        step()
        store(done=t == T - 1, use_mask=True)
    assert torch.equal(g._t["partner"], g2._t["partner"]) and torch.equal(g._t["n_groups"], g2._t["n_groups"])
    for k in ("data_buf", "reward", "obs", "metrics", "mec_q", "gain"):
        assert torch.equal(env.tensors[k], env2.tensors[k]), k
    assert buf.mem_cntr == buf2.mem_cntr == T * E
    for k in buf._ARRAYS:
        assert torch.equal(getattr(buf, k), getattr(buf2, k)), k
    # and the unbound store_batch(policy_out=...) form
    buf3 = VecReplayBuffer(2 * E, 5, V + 2, V, device=DEV)
    buf3.store_batch(state_old, None, env.tensors["metrics"], env.tensors["reward"], env.tensors["obs"], False, mask,
                     policy_out=(power[-1], probs[-1]))
    _, _, a_store = marshal_actions(power[-1], probs[-1], float(env.cpu_share_floor))
    assert torch.equal(buf3.action_memory[:E], a_store)
    with pytest.raises(ValueError):
        buf3.store_batch(state_old, None, env.tensors["metrics"], env.tensors["reward"], env.tensors["obs"])


@pytest.mark.parametrize("V,M,fused", [(8, 64, True), (8, 36, True), (4, 16, True), (16, 64, True), (8, 64, False), (16, 50, False),
                                       (4, 30, False)])
def test_step_with_the_transition_store_fused_in(V, M, fused):
    """`bind_step_store` (risvec_step_ring): ONE launch = step() + this step's E transitions appended to the ring
    (marl_train_bcd.py:1601-1611, 1776-1799; buffer.py:16-25).  Env tensors and all seven ring arrays must equal the
    two-launch form -- `bind_step(policy_action=True)` then `store_batch(state = the observation before the step,
    policy_out = (power_raw, probs))` -- bit for bit, through a ring wrap, with and without the NOMA mask, with the
    terminal flag, from a reset (the first `state` is the observation assembled from the reset state)."""
    from ris_vec_marl_amd import VecEnviron, VecReplayBuffer, apply_yaml_config, reference_lanes
    from ris_vec_marl_amd import _native as N
    E, T = 777, 7
    L = reference_lanes()
    gen = torch.Generator(device=DEV); gen.manual_seed(11 + V + M)
    power = [torch.rand(E, V, 2, device=DEV, generator=gen) * 2.4 - 1.2 for _ in range(T)]
    probs = [torch.softmax(torch.randn(E, V, V, device=DEV, generator=gen), -1) for _ in range(T)]
    mask = (torch.rand(E, V, V, device=DEV, generator=gen) < 0.6).to(torch.uint8)
    partner = torch.full((E, V), -1, dtype=torch.int32, device=DEV); partner[:, 0] = 1; partner[:, 1] = (1 << 16)
    ng = torch.full((E,), V - 1, dtype=torch.int32, device=DEV)

    def build():
        env = VecEnviron(L["down_lanes"], L["up_lanes"], L["left_lanes"], L["right_lanes"], 400, 400, V, M, 3, n_envs=E,
                         device=DEV, seed=21)
        apply_yaml_config(env, None)
        env.make_new_game(); env.renew_positions(); env.compute_parms(); env.Random_phase(); env.update_channel_gains()
        return env, VecReplayBuffer(int(2.5 * E), 5, V + 2, V, device=DEV)      # wraps during step 3

    # (a) two launches per step
    env, buf = build()
    pw, pr = torch.empty(E, V, 2, device=DEV), torch.empty(E, V, V, device=DEV)
    step = env.bind_step(pw, partner, ng, None, fused=fused, policy_action=True, power_w=False)
    for t in range(T):
        pw.copy_(power[t]); pr.copy_(probs[t])
        before = env.observe().clone()
        step()
        buf.store_batch(before, None, env.tensors["metrics"], env.tensors["reward"], env.tensors["obs"], t == T - 1,
                        mask if t % 2 == 0 else None, policy_out=(pw, pr))
    # (b) one launch per step
    env2, buf2 = build()
    pw2, pr2 = torch.empty(E, V, 2, device=DEV), torch.empty(E, V, V, device=DEV)
    both = env2.bind_step_store(buf2, pw2, partner, ng, pr2, mask, fused=fused)
    for t in range(T):
        pw2.copy_(power[t]); pr2.copy_(probs[t])
        both(done=t == T - 1, use_mask=t % 2 == 0)
        assert N.last_kernel().endswith("MarlCore+ring>") if fused else N.last_kernel() == "k_step<%d,RING>" % V, N.last_kernel()
    assert buf.mem_cntr == buf2.mem_cntr == T * E and env._steps == env2._steps == T
    for k in ("data_buf", "reward", "obs", "metrics", "mec_q", "gain", "rate", "data_t", "data_p", "over_power"):
        assert torch.equal(env.tensors[k], env2.tensors[k]), k
    for k in buf._ARRAYS:
        assert torch.equal(getattr(buf, k), getattr(buf2, k)), k
    # shapes without a fused-gains form are refused, not silently run without the store
    if fused:
        env3 = VecEnviron(L["down_lanes"], L["up_lanes"], L["left_lanes"], L["right_lanes"], 400, 400, V, 22, 3, n_envs=E,
                          device=DEV, seed=21)
        env3.make_new_game(); env3.compute_parms()
        bad = env3.bind_step_store(buf2, pw2, partner, ng, pr2, None, fused=True)
        with pytest.raises(N.RisVecError):
            bad()
