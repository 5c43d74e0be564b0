"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol
include/risvec.h declares, host logic (YAML key map, NOMA-group encoding, facade
surface, sharding) behaves like the reference's driver expects, and the product
fails loudly without a HIP device."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

import ris_vec_marl_amd as rv
from ris_vec_marl_amd import _native as N
from oracle import risvec_oracle as orc
from oracle import noma_oracle as orc_noma

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_YAML_EFFECTIVE = dict(w_d=1.0, w_e=1.0, rate=1.0, f_local_max=3.0e9, cycles_per_bit=300.0, bandwidth=5.0,
                          P_max=2.0, power_scale=0.7, cpu_share_floor=0.10, f_edge_max=2.0e9, k=1e-28,
                          qos_enable=True, R_min_bpsHz=0.15, D_max_s=0.12, qos_penalty=1.5)
# the shipped config.yaml keys that reach the env (data, not reference source)
SHIPPED_YAML = dict(mec=dict(f_local_max=3.0e9, cycles_per_bit=300), phy=dict(bandwidth_MHz=5, P_max=2.0),
                    env=dict(rate=1), reward=dict(sample=False, w_d_fixed=1.0, w_e_fixed=1.0, norm_beta=0.9),
                    qos_enable=True, qos_penalty=1.5, env_refresh_every=5, seed=42, entropy_scale=0.5)


def make_facade(V=8, M=40):
    L = rv.reference_lanes()
    return rv.Environ(L["down_lanes"], L["up_lanes"], L["left_lanes"], L["right_lanes"], 400, 400, V, M, 3)


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "risvec.h")).read()
    declared = set(re.findall(r"\b(risvec_[a-z0-9_]+)\s*\(", header))
    declared.discard("risvec_stream_t")
    assert declared == set(N.EXPORTS), declared ^ set(N.EXPORTS)
    lib = N.load()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.risvec_abi_version() == N.ABI_VERSION
    m = re.search(r"#define RISVEC_ABI_VERSION (\d+)", header)
    assert int(m.group(1)) == N.ABI_VERSION


def test_struct_layout_and_default_params_match_python():
    lib = N.load()
    p = N.RisVecParams()
    lib.risvec_default_params(C.byref(p))
    assert p.struct_bytes == C.sizeof(N.RisVecParams)
    L = rv.reference_lanes()
    q = rv.EnvParams().to_c(L, 400, 400)
    for name, _ in N.RisVecParams._fields_:
        a, b = getattr(p, name), getattr(q, name)
        if hasattr(a, "__len__"):
            np.testing.assert_allclose(list(a), list(b), rtol=2e-6, err_msg=name)
        else:
            assert a == pytest.approx(b, rel=1e-6), name
    # product table == oracle table, bit for bit (the GPU arrivals test relies on it)
    for lam in (1.0, 3.0, 0.25, 7.5):
        assert np.array_equal(rv.poisson_cdf_table(lam), orc.poisson_cdf_table(lam))
    with pytest.raises(ValueError):
        rv.poisson_cdf_table(40.0)


def test_api_rejects_bad_arguments_without_touching_a_gpu():
    lib = N.load()
    p = N.RisVecParams()
    lib.risvec_default_params(C.byref(p))
    s = N.RisVecState()
    assert lib.risvec_step(C.byref(s), C.byref(p), None, None, None, None, 0, 0, 0, None) == N.ERR_ARG
    s.abi_version, s.struct_bytes = N.ABI_VERSION, C.sizeof(N.RisVecState)
    s.n_envs, s.n_veh, s.n_ris, s.control_bit = 4, 100, 16, 3
    assert lib.risvec_gain(C.byref(s), C.byref(p), None) == N.ERR_SHAPE
    assert b"n_veh" in lib.risvec_last_error()
    s.n_veh = 8
    assert lib.risvec_gain(C.byref(s), C.byref(p), None) == N.ERR_ARG          # NULL device pointers
    assert b"NULL" in lib.risvec_last_error()
    with pytest.raises(ValueError):
        N.check(N.ERR_ARG)
    # the two T-step entry points validate before they launch: n_steps, flags, NULL inputs
    for fn in (lib.risvec_step_fused_multi, lib.risvec_step_multi):
        assert fn(C.byref(s), None, 4, None, None, None, None, 0, 0, 0, None, None) == N.ERR_ARG       # params NULL
        assert fn(C.byref(s), C.byref(p), 0, None, None, None, None, 0, 0, 0, None, None) == N.ERR_ARG
        assert b"n_steps" in lib.risvec_last_error()
        assert fn(C.byref(s), C.byref(p), 4, None, None, None, None, 0, 0, N.STEP_REUSE_IDX, None, None) == N.ERR_ARG
        assert fn(C.byref(s), C.byref(p), 4, None, None, None, None, 0, 0, 0, None, None) == N.ERR_ARG  # NULL actions
        assert b"NULL" in lib.risvec_last_error()


def test_yaml_key_map_matches_reference_effective_values():
    env = make_facade()
    # class defaults first (Environment.py:57-190)
    assert (env.w_d, env.w_e, env.rate, env.P_max, env.bandwidth) == (0.5, 3.0, 3, 1.0, 1.0)
    assert env.noise_power == pytest.approx(3.981071705534986e-15, rel=1e-12)
    rv.apply_yaml_config(env, SHIPPED_YAML)
    for k, v in REF_YAML_EFFECTIVE.items():
        assert getattr(env, k) == pytest.approx(v), k
    assert env.noise_power == pytest.approx(1.990535852767493e-14, rel=1e-12)
    assert env.bandwidth_hz == 5e6 and env.channel_model == "free"
    # without a YAML the driver's Config defaults apply (marl_train_bcd.py:426-427, 505-508, 750-753)
    env2 = make_facade()
    rv.apply_yaml_config(env2, None)
    assert (env2.w_d, env2.w_e, env2.qos_penalty, env2.R_min_bpsHz, env2.D_max_s) == (1.0, 2.0, 5.0, 0.15, 0.12)
    # oracle's notion of the effective parameters agrees with the product's
    p = orc.OracleParams.yaml_effective()
    for a, b in (("bandwidth", "bandwidth"), ("noise_power", "noise_power"), ("P_max", "P_max"), ("w_d", "w_d"),
                 ("w_e", "w_e"), ("rate", "rate"), ("f_local_max", "f_local_max"), ("cycles_per_bit", "cycles_per_bit"),
                 ("R_min_bpsHz", "R_min_bpsHz"), ("D_max_s", "D_max_s"), ("qos_penalty", "qos_penalty")):
        assert getattr(p, a) == pytest.approx(getattr(env, b)), a


def test_params_reach_the_c_struct():
    env = make_facade()
    rv.apply_yaml_config(env, SHIPPED_YAML)
    c = env._vec._p()
    assert c.bandwidth_mhz == 5.0 and c.p_max == 2.0 and c.cycles_per_bit == 300.0
    assert c.noise_power == pytest.approx(1.990535852767493e-14, rel=1e-6)
    v0 = env.params.version
    env.w_d = 0.25
    assert env.params.version > v0 and env._vec._p().w_d == 0.25
    assert list(c.lanes_up)[:4] == rv.reference_lanes()["up_lanes"]


def test_encode_noma_groups():
    groups = [[0, 3], [5, 2], [1], [4], [6, 7, 1]]           # the last group is ignored but counted
    partner, ng = rv.encode_noma_groups([groups], 8)
    want, n = orc.encode_groups(groups[:-1], 8)
    assert ng[0] == 5 and n == 4
    assert np.array_equal(partner[0], want)
    assert partner[0, 0] == 3 and partner[0, 3] == 0 + (1 << 16) and partner[0, 1] == -1 and partner[0, 6] == -2
    # a vehicle listed twice: the last group that lists it decides (Environment.py:344-369; pinned by
    # tests/golden/facade_groups_8.npz), its earlier partner keeps the pair entry
    p1, n1 = rv.encode_noma_groups([[[0, 1], [1]]], 8)
    assert p1[0, 0] == 1 and p1[0, 1] == -1 and n1[0] == 2
    with pytest.raises(ValueError):
        rv.encode_noma_groups([[[0, 9]]], 8)
    p2, n2 = rv.encode_noma_groups([[], [[2]]], 4)
    assert n2.tolist() == [0, 1] and (p2[0] == -2).all()


def test_facade_surface_matches_reference_inventory():
    """names a driver may touch on `Environ`."""
    env = make_facade()
    methods = ["Random_phase", "add_new_vehicles", "add_new_vehicles_by_number", "compute_data_rate", "compute_parms",
               "get_channel_gains", "get_next_phase", "get_path_loss", "get_shadowing", "localProcRev", "make_new_game",
               "optimize_compute_objective_function", "optimize_phase_shift", "renew_positions", "step",
               "update_channel_gains"]
    for m in methods:
        assert callable(getattr(env, m)), m
    host_attrs = ["down_lanes", "up_lanes", "left_lanes", "right_lanes", "width", "height", "n_veh", "M", "control_bit",
                  "possible_angles", "distance_B_R", "angle_B_R", "channel_model", "fc_GHz", "bandwidth", "bandwidth_hz",
                  "N0_dBm_per_Hz", "N0_W_per_Hz", "noise_power", "P_max", "shadow_std_los", "shadow_std_nlos",
                  "rician_K_dB", "Decorrelation_distance", "V2I_Shadowing", "V2I_pathloss", "V2I_channels_abs",
                  "delta_distance", "sig2_dB", "sig2", "bsAntGain", "bsNoiseFigure", "vehAntGain", "vehNoiseFigure",
                  "time_slow", "time_fast", "k", "L", "f_local_max", "f_edge_max", "cycles_per_bit", "cpu_share_floor",
                  "power_scale", "qos_enable", "R_min_bpsHz", "D_max_s", "qos_penalty", "w_d", "w_e", "reward_clip",
                  "sample_weights", "w_d_range", "w_e_range", "w_fair_range", "w_fair", "reward_scale",
                  "reward_norm_beta", "delay_mean", "delay_var", "energy_mean", "energy_var", "rate", "data_buf_size",
                  "data_r", "phase_R", "elements_phase_shift_real", "last_off_kbit_sum", "last_local_kbit_sum",
                  "last_mec_queue_cycles"]
    for a in host_attrs:
        assert hasattr(env, a), a
    assert env.distance_B_R == pytest.approx(311.1269837220809) and env.angle_B_R == pytest.approx(0.7071067811865475)
    assert len(env.possible_angles) == 8 and env.phase_R.shape == (40,)
    np.testing.assert_allclose(env.phase_R, orc.phase_R(40), atol=1e-15)
    # device-backed names exist on the class (they need a GPU to be read)
    for a in ["DataBuf", "data_t", "data_p", "over_data", "vehicle_rate", "channel_gains", "distances_R_i",
              "angles_R_i", "mec_queue_cycles", "elements_phase_shift_complex", "phases_R_i", "vehicles"]:
        assert isinstance(getattr(type(env), a), property), a
    # created lazily by step() in the reference: absent until then (driver uses getattr(..., None))
    assert getattr(env, "last_delay_mean", None) is None
    assert getattr(env, "last_qos_violation", None) is None


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_cpu_fallback():
    env = make_facade()
    for call in (env.make_new_game, env.renew_positions, env.compute_parms, env.optimize_phase_shift,
                 env.update_channel_gains, lambda: env.step(np.zeros((2, 8)), [[0]]), lambda: env.DataBuf):
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            call()
    for dead in (lambda: env.get_path_loss([0, 0]), lambda: env.localProcRev(1.0)):
        with pytest.raises(NotImplementedError):
            dead()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "ris_vec_marl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
                assert "/root/reference" not in src, f


def test_shard_range():
    for n, w in ((65536, 8), (10, 3), (7, 8), (32768, 1)):
        spans = [rv.dist.shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == n
        for (s0, c0), (s1, _) in zip(spans, spans[1:]):
            assert s0 + c0 == s1
        assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    with pytest.raises(ValueError):
        rv.dist.shard_range(8, 8, 8)


def test_noma_and_replay_abi_without_a_gpu():
    """f2 / f3 entry points: struct layouts agree between ctypes and the library (the C defaults land
    in the right Python fields), and bad arguments are rejected before anything is launched."""
    lib = N.load()
    for V in (4, 8, 16):
        p = N.RisVecNomaParams()
        lib.risvec_noma_default_params(C.byref(p), V)
        q = rv.NomaConfig(V).to_c(noise_power=10 ** (-174 / 10) / 1000 * 1e6, P_max=1.0)
        for name, _ in N.RisVecNomaParams._fields_:
            a, b = getattr(p, name), getattr(q, name)
            assert a == pytest.approx(b, rel=1e-6) or (a == b), (V, name, a, b)
    ns = N.RisVecNomaState()
    p = rv.NomaConfig(8).to_c(1e-14, 1.0)
    assert lib.risvec_noma_begin_episode(None, None) == N.ERR_ARG
    ns.n_envs, ns.n_veh = 4, 17
    assert lib.risvec_noma_mask(C.byref(ns), None, None, 0.5, 3, None) == N.ERR_SHAPE
    assert b"n_veh" in lib.risvec_last_error()
    ns.n_veh = 8
    assert lib.risvec_noma_mask(C.byref(ns), None, None, 0.5, 3, None) == N.ERR_ARG        # NULL gain
    assert lib.risvec_noma_group(C.byref(ns), C.byref(p), None, None, None, 0, 3, None, None, 1, 0, None, 0, 0, None,
                                 None) == N.ERR_ARG
    assert lib.risvec_noma_flush(C.byref(ns), None, None) == N.ERR_ARG
    # scratch: none up to 8 vehicles; beyond, the list of envs the first launch leaves to the second (16 B + an int per env)
    assert lib.risvec_noma_scratch_bytes(32768, 8) == 0
    assert lib.risvec_noma_scratch_bytes(32768, 16) == 16 + 4 * 32768 + 240
    assert lib.risvec_noma_scratch_bytes(17, 9) == 256
    assert lib.risvec_noma_scratch_bytes(0, 16) == 0 and lib.risvec_noma_scratch_bytes(8, 17) == 0
    rb = N.RisVecReplay()
    assert lib.risvec_replay_store(C.byref(rb), 0, 1, None, None, None, 1, None, None, None, 0, None, None, None) \
        == N.ERR_SHAPE
    rb.n_agents, rb.input_shape, rb.n_actions, rb.mem_size = 8, 5, 10, 100
    assert lib.risvec_replay_store(C.byref(rb), 0, 1, None, None, None, 1, None, None, None, 0, None, None, None) \
        == N.ERR_ARG
    assert lib.risvec_replay_sample(C.byref(rb), 0, 4, None, 0, 0, None, None, None, None, None, None, None, None,
                                    None) == N.ERR_ARG
    assert lib.risvec_marshal_actions(0, 8, None, None, 0.1, None, None, None, None) == N.ERR_SHAPE
    assert lib.risvec_marshal_actions(4, 8, None, None, 0.1, None, None, None, None) == N.ERR_ARG
    with pytest.raises(NotImplementedError):
        c = rv.NomaConfig(8); c.use_mwm_primary = False; c.to_c(1e-14, 1.0)


def test_noma_config_reads_the_drivers_yaml_keys():
    """NomaConfig.apply_yaml takes the same keys from the same places as marl_train_bcd.py:575,
    639-660, 716-741; values below are the shipped config.yaml's (its lines 39-44, 53, 78-83, 116-120, 144-148)."""
    y = {"min_pair_target": 3, "use_mwm_primary": True, "mwm_accept_quantile": 0.10, "mwm_backoff_rounds": 3,
         "mwm_accept_q_step": 0.05, "qos_enable": True, "freeze_group_in_episode": True, "freeze_recalc_every": 0,
         "freeze_unstick_prob": 0.0, "freeze_reward_drop_ratio": 0.05,
         "reward": {"mask_enable": True, "mask_topk_start": 7, "mask_topk_end": 7, "mask_tau_q_start": 0.10,
                    "mask_tau_q_end": 0.25, "pairing_threshold_quantile": 0.25}}
    c = rv.NomaConfig(8).apply_yaml(y)
    o = orc_noma.NomaParams.yaml_effective(8)
    for k in ("min_pair_target", "mwm_accept_quantile", "mwm_backoff_rounds", "mwm_accept_q_step", "qos_enable",
              "qos_R_min_bpsHz", "mask_topk_start", "mask_topk_end", "mask_tau_q_start", "mask_tau_q_end",
              "pairing_threshold_quantile", "freeze_group_in_episode", "freeze_recalc_every"):
        assert getattr(c, k) == getattr(o, k), k
    assert c.mask_schedule(50) == orc_noma.mask_schedule(o, 50, 8)
    assert rv.anneal_topk(37, 8, 7, 4, 200) == orc_noma.anneal_topk(37, 8, 7, 4, 200)


def test_policy_abi_without_a_gpu():
    lib = N.load()
    assert lib.risvec_policy_sample(0, 8, 0, None, None, None, None, None, None, 0, 0, 0.1, None, None, None, None, None,
                                    None, None) == N.ERR_SHAPE
    assert lib.risvec_policy_sample(4, 8, 0, None, None, None, None, None, None, 0, 0, 0.1, None, None, None, None, None,
                                    None, None) == N.ERR_ARG
    assert lib.risvec_policy_layer1(4, 8, 5, 4096, None, None, None, None, None, None, None) == N.ERR_SHAPE
    assert b"f1" in lib.risvec_last_error()
    assert lib.risvec_policy_layer1(4, 8, 5, 512, None, None, None, None, None, None, None) == N.ERR_ARG
    assert lib.risvec_policy_heads(4, 8, 256, 12, None, None, None, None, None, None, None, None) == N.ERR_ARG
    assert lib.risvec_policy_heads(4, 8, 1024, 68, None, None, None, None, None, None, None, None) == N.ERR_SHAPE   # > 64 KB of LDS


def test_episode_abi_without_a_gpu():
    lib = N.load()
    assert lib.risvec_episode_partial_rows(0) == 0 and lib.risvec_episode_partial_rows(1) == 1
    assert lib.risvec_episode_partial_rows(256) == 1 and lib.risvec_episode_partial_rows(32768 + 1) == 129
    assert lib.risvec_episode_clear(0, 8, None, None) == N.ERR_SHAPE
    assert lib.risvec_episode_clear(4, 65, None, None) == N.ERR_SHAPE
    assert lib.risvec_episode_clear(4, 8, None, None) == N.ERR_ARG and b"acc" in lib.risvec_last_error()
    assert lib.risvec_episode_clear(1 << 27, 8, None, None) == N.ERR_SHAPE          # E * (17 + V) >= 2^31
    assert lib.risvec_episode_accumulate(4, 8, None, None, None, 5.0, None, None) == N.ERR_ARG
    assert lib.risvec_episode_accumulate(4, 8, 16, 16, None, 5.0, 8, None) == N.ERR_ARG      # acc not 16-byte aligned
    assert lib.risvec_episode_accumulate(4, 8, 16, 16, None, -1.0, 16, None) == N.ERR_ARG and b"user_clip" in lib.risvec_last_error()
    assert lib.risvec_episode_accumulate(4, 8, 16, 16, None, float("nan"), 16, None) == N.ERR_ARG
    assert lib.risvec_episode_summary(4, 8, 0, 16, 16, None, 16, 16, None) == N.ERR_ARG and b"n_steps" in lib.risvec_last_error()
    assert lib.risvec_episode_summary(4, 8, 3, 16, 16, None, None, 16, None) == N.ERR_ARG


def test_policy_split16_abi_without_a_gpu():
    lib = N.load()
    assert lib.risvec_policy_layer1_split16(4, 8, 5, 510, None, None, None, None, None, None, None) == N.ERR_SHAPE
    assert lib.risvec_policy_layer1_split16(4, 8, 9, 512, None, None, None, None, None, None, None) == N.ERR_SHAPE
    assert lib.risvec_policy_layer1_split16(4, 8, 5, 512, None, None, None, None, None, None, None) == N.ERR_ARG
