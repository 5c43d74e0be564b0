"""GPU parity tests of the batched choose_action: `BatchedPolicy` (torch.bmm for the
three GEMMs + the hand-written sampling / marshalling epilogue risvec_policy_sample) against the vectors
captured from the reference's own PolicyNetwork objects and against oracle/policy_oracle.py.

Tolerance: the reference computes in float32 on the CPU; here float32 on the GPU with another
accumulation order -> 2e-5 absolute on tanh / softmax outputs (values in [-1,1] / [0,1]); one-hot exact
wherever the top two probabilities differ by more than 1e-4; the marshalled outputs must equal what
`marshal_actions` makes of the same power / probs bit for bit."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from oracle import policy_oracle as PO  # noqa: E402  (checker)
from oracle import risvec_oracle as orc  # noqa: E402  (checker)

GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def agent_weights(d, a):
    pre = "a%d." % a
    return {k[len(pre):]: d[k] for k in d.files if k.startswith(pre)}


@pytest.mark.parametrize("gemm", ["default", "fp16x3", "fp32"])
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "policy_*.npz"))), ids=os.path.basename)
def test_vs_golden(path, gemm):
    from ris_vec_marl_amd import BatchedPolicy, marshal_actions
    d = np.load(path)
    V, B = int(d["V"]), int(d["B"])
    pol = BatchedPolicy(V, 5, int(d["fc1"]), int(d["fc2"]), device=DEV, gemm=None if gemm == "default" else gemm)
    assert pol.gemm == ("fp16x3" if gemm == "default" else gemm)      # 48/32 and 40/24 hidden units: no fused build
    for a in range(V):
        pol.load_agent_state_dict(a, agent_weights(d, a))
        pol.tau[a] = float(d["tau"][a])
        pol.gumbel_hard[a] = int(d["hard"][a])
    obs = np.stack([d["state"][a] for a in range(V)], 1)                      # [B,V,5]
    mask = np.stack([d["mask"][a] if d["has_mask"][a] else np.ones((B, V), np.float32) for a in range(V)], 1)
    eps = np.stack([d["eps"][a] for a in range(V)], 1)
    expo = np.stack([d["expo"][a] for a in range(V)], 1)
    heads = pol.forward_heads(T(obs)).cpu().numpy()                            # [V,B,4+V]
    for a in range(V):
        np.testing.assert_allclose(heads[a][:, 0:2], d["mu"][a], atol=5e-6)
        np.testing.assert_allclose(np.clip(heads[a][:, 2:4], -20, 2), d["log_std"][a], atol=5e-6)
        np.testing.assert_allclose(heads[a][:, 4:], d["logits"][a], atol=5e-6)
    power, probs, onehot, a_env, p01, a_store = pol.choose_action(T(obs), T(mask), T(eps), T(expo), cpu_share_floor=0.1)
    for a in range(V):
        np.testing.assert_allclose(power.cpu().numpy()[:, a], d["power"][a], atol=2e-5)
        w = agent_weights(d, a)
        soft = PO.choose_action(w, d["state"][a], d["mask"][a] if d["has_mask"][a] else None, float(d["tau"][a]),
                                d["eps"][a], d["expo"][a])[1]
        clear = PO.top2_gap(soft) > 1e-4
        assert np.array_equal(onehot.cpu().numpy()[:, a][clear], d["onehot"][a][clear])
        if d["hard"][a]:                      # straight-through: the one-hot up to one float32 rounding
            np.testing.assert_allclose(probs.cpu().numpy()[:, a][clear], d["probs"][a][clear], atol=2e-7)
        else:
            np.testing.assert_allclose(probs.cpu().numpy()[:, a], d["probs"][a], atol=2e-5)
    m_env, m_p01, m_store = marshal_actions(power, probs, 0.1)               # fused marshalling == separate launch
    assert torch.equal(a_env, m_env) and torch.equal(p01, m_p01) and torch.equal(a_store, m_store)
    mine = (torch.empty_like(a_env), torch.empty_like(p01), torch.empty_like(a_store))     # caller-owned outputs
    pol._calls -= 1
    got = pol.choose_action(T(obs), T(mask), T(eps), T(expo), cpu_share_floor=0.1, out=mine)
    assert got[3] is mine[0] and all(torch.equal(x, y) for x, y in zip(mine, (a_env, p01, a_store)))
    with pytest.raises(ValueError):
        pol.choose_action(T(obs), T(mask), T(eps), T(expo), cpu_share_floor=0.1, out=(mine[0], mine[1], mine[2].double()))


@pytest.mark.parametrize("shape", [(8, 512, 256, 1024), (5, 64, 32, 300), (16, 128, 64, 300), (11, 32, 32, 200), (1, 16, 8, 50)])
def test_reference_sized_networks_vs_oracle(shape):
    """The driver's sizes (512 / 256 hidden, config A_fc1_dims / A_fc2_dims), 8 agents, E = 1 024 -- and
    agent counts that are not powers of two (padding lanes in the row-per-lane-group sampling kernel) --
    random weights in the reference's init ranges: device float32 vs the float64 oracle."""
    from ris_vec_marl_amd import BatchedPolicy
    V, F1, F2, E = shape
    rng = np.random.default_rng(3)
    pol = BatchedPolicy(V, 5, F1, F2, device=DEV, seed=11)
    with torch.no_grad():                       # heads start at +-0.003: widen them so the outputs are not all ~0
        pol.Wh.mul_(60.0)
    obs = rng.uniform(0, 1.2, (E, V, 5)).astype(np.float32)
    mask = (rng.uniform(size=(E, V, V)) < 0.7).astype(np.uint8)
    eps = rng.normal(size=(E, V, 2)).astype(np.float32)
    expo = rng.exponential(size=(E, V, V)).astype(np.float32)
    pol.set_tau(1.0)
    power, probs, onehot = pol.choose_action(T(obs), T(mask), T(eps), T(expo))
    sd = {k: v.numpy() for k, v in pol.state_dict().items()}
    for a in range(V):
        w = {"fc1.weight": sd["W1"][a].T, "fc1.bias": sd["b1"][a, 0], "fc2.weight": sd["W2"][a].T, "fc2.bias": sd["b2"][a, 0],
             "bn1.weight": sd["ln1_w"][a, 0], "bn1.bias": sd["ln1_b"][a, 0], "bn2.weight": sd["ln2_w"][a, 0],
             "bn2.bias": sd["ln2_b"][a, 0], "mu.weight": sd["Wh"][a][:, 0:2].T, "mu.bias": sd["bh"][a, 0, 0:2],
             "log_std.weight": sd["Wh"][a][:, 2:4].T, "log_std.bias": sd["bh"][a, 0, 2:4],
             "intent_logits.weight": sd["Wh"][a][:, 4:].T, "intent_logits.bias": sd["bh"][a, 0, 4:]}
        po, yo, oh = PO.choose_action(w, obs[:, a], mask[:, a].astype(np.float64), 1.0, eps[:, a], expo[:, a])
        np.testing.assert_allclose(power.cpu().numpy()[:, a], po, atol=3e-5)
        np.testing.assert_allclose(probs.cpu().numpy()[:, a], yo, atol=3e-5)
        clear = PO.top2_gap(yo) > 1e-3
        assert clear.mean() > 0.9 and np.array_equal(onehot.cpu().numpy()[:, a][clear], oh[clear])


def test_full_size_properties_and_philox():
    """E = 32 768 x 8 agents, production draws: probabilities sum to 1, blocked partners get none, the
    one-hot marks the arg-max, tanh range; the Philox draws are the documented ones (site 9 / 10)."""
    from ris_vec_marl_amd import BatchedPolicy
    V, E = 8, 32768
    rng = np.random.default_rng(5)
    pol = BatchedPolicy(V, 5, 512, 256, device=DEV, seed=77, env_offset=4096)
    with torch.no_grad():
        pol.Wh.mul_(100.0)
    obs = T(rng.uniform(0, 1.2, (E, V, 5)).astype(np.float32))
    mask = (rng.uniform(size=(E, V, V)) < 0.7).astype(np.uint8)
    mask[:100] = 0                                              # all-zero rows are opened up
    power, probs, onehot = pol.choose_action(obs, T(mask))
    p, y, oh = power.cpu().numpy(), probs.cpu().numpy(), onehot.cpu().numpy()
    assert np.all(np.abs(p) <= 1.0) and np.isfinite(y).all()
    np.testing.assert_allclose(y.sum(-1), 1.0, atol=1e-5)
    blocked = (mask == 0) & (mask.sum(-1, keepdims=True) > 0)
    assert np.all(y[blocked] == 0.0)
    assert np.all(y[:100].min(-1) > 0)
    assert np.array_equal(oh.argmax(-1), y.argmax(-1)) and np.all(oh.sum(-1) == 1.0)
    # the draws: eps = Box-Muller of Philox(seed; env, agent, call, 9); recompute power from heads
    heads = pol.forward_heads(obs).cpu().numpy().astype(np.float64)      # [V,E,4+V]
    e_ids = np.arange(4096, 4096 + E, dtype=np.uint64)
    for a in (0, 5):
        x = orc.philox4x32(e_ids, np.full(E, a, np.uint64), np.full(E, 1, np.uint64), np.full(E, 9, np.uint64), 77)
        u1 = ((x[0] >> 8).astype(np.float64) + 1.0) * 2.0 ** -24
        u2 = (x[1] >> 8).astype(np.float64) * 2.0 ** -24
        r = np.sqrt(-2.0 * np.log(u1))
        n0, n1 = r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)
        ls = np.clip(heads[a][:, 2:4], -20, 2)
        np.testing.assert_allclose(p[:, a, 0], np.tanh(n0 * np.exp(ls[:, 0]) + heads[a][:, 0]), atol=2e-4)
        np.testing.assert_allclose(p[:, a, 1], np.tanh(n1 * np.exp(ls[:, 1]) + heads[a][:, 1]), atol=2e-4)


@pytest.mark.parametrize("gemm", ["fused", "fp16x3", "fp32"])
@pytest.mark.parametrize("dims", [(8, 512, 256), (4, 40, 24), (16, 128, 64), (3, 1000, 70), (16, 512, 256), (5, 96, 128),
                                  (8, 1024, 256), (2, 32, 128)])
def test_handwritten_forward_vs_library_forward(dims, gemm):
    """k_policy_layer1 / k_policy_heads (+ the rocBLAS fc2 GEMM) against the same forward done with library
    kernels only (torch.bmm + layer_norm), fp32 both: 1e-5 relative to the row scale."""
    from ris_vec_marl_amd import BatchedPolicy
    V, F1, F2 = dims
    E = 777
    if gemm == "fused" and not (F1 % 32 == 0 and F2 in (128, 256)):
        with pytest.raises(ValueError):
            BatchedPolicy(V, 5, F1, F2, device=DEV, seed=V, gemm=gemm)
        return
    pol = BatchedPolicy(V, 5, F1, F2, device=DEV, seed=V, gemm=gemm)
    with torch.no_grad():
        pol.Wh.mul_(50.0)
        pol.ln1_w.uniform_(0.5, 1.5); pol.ln1_b.uniform_(-0.2, 0.2); pol.ln2_w.uniform_(0.5, 1.5); pol.ln2_b.uniform_(-0.2, 0.2)
    obs = torch.rand(E, V, 5, device=DEV) * 1.2
    a, b = pol.forward_heads(obs), pol.forward_heads_torch(obs)
    scale = b.abs().amax(dim=-1, keepdim=True).clamp_min(1e-3)
    assert float(((a - b).abs() / scale).max()) < 2e-5


def test_split_fp16_gemm_has_float32_accuracy():
    """The fc1 x fc2 product as ONE float16 GEMM over the split operands (risvec_policy_layer1_split16 x
    the stacked fc2 weight) against the float64 product of the same float32 hidden rows: its error is
    of the order of the float32 library GEMM's, far below the 1e-5 bar; stale-weight cache and the
    shapes the split form refuses.  (Weights below the float16 normal range, |w| < 6e-5, keep an ABSOLUTE
    error of 2^-25 each -- irrelevant next to the other columns of the LayerNorm row they feed, which
    is why the bound here is checked on weights of realistic size.)"""
    from ris_vec_marl_amd import BatchedPolicy
    from ris_vec_marl_amd import _native as N
    V, F1, F2, E = 8, 512, 256, 2048
    pol = BatchedPolicy(V, 5, F1, F2, device=DEV, seed=5, gemm="fp16x3")
    with torch.no_grad():
        pol.ln1_w.uniform_(0.5, 3.0); pol.ln1_b.uniform_(-0.5, 0.5)
        pol.W2.mul_(torch.logspace(-1.5, 1, F2, device=DEV))         # columns from small (2e-3) to large (0.6) weights
    obs = torch.rand(E, V, 5, device=DEV) * 1.2
    lib, st = N.load(), torch.cuda.current_stream().cuda_stream
    h32 = torch.empty(V, E, F1, device=DEV)
    N.check(lib.risvec_policy_layer1(E, V, 5, F1, obs.data_ptr(), pol.W1.data_ptr(), pol.b1.data_ptr(), pol.ln1_w.data_ptr(),
                                     pol.ln1_b.data_ptr(), h32.data_ptr(), st))
    h16 = torch.empty(V, E, 3 * F1, dtype=torch.float16, device=DEV)
    N.check(lib.risvec_policy_layer1_split16(E, V, 5, F1, obs.data_ptr(), pol.W1.data_ptr(), pol.b1.data_ptr(),
                                             pol.ln1_w.data_ptr(), pol.ln1_b.data_ptr(), h16.data_ptr(), st))
    # the three K-blocks are what the header says, exactly
    hi = h32.to(torch.float16)
    assert torch.equal(h16[..., :F1], hi)
    assert torch.equal(h16[..., F1:2 * F1], (hi.float() * 0.03125).to(torch.float16))
    assert torch.equal(h16[..., 2 * F1:], ((h32 - hi.float()) * 64.0).to(torch.float16))
    exact = torch.bmm(h32.double(), pol.W2.double())
    split = torch.bmm(h16, pol._split_w2(), out_dtype=torch.float32).double()
    plain = torch.bmm(h32, pol.W2).double()
    scale = (h32.double().abs() @ pol.W2.double().abs()).clamp_min(1e-30)     # sum |a||b|: the natural error scale
    e_split, e_plain = float(((split - exact).abs() / scale).max()), float(((plain - exact).abs() / scale).max())
    assert e_split < 2e-6 and e_split < 8 * max(e_plain, 1e-7), (e_split, e_plain)
    # the split weight follows in-place updates of W2
    before = pol.forward_heads(obs).clone()
    with torch.no_grad():
        pol.W2.mul_(1.5)
    after = pol.forward_heads(obs)
    assert not torch.equal(before, after)
    ref = pol.forward_heads_torch(obs)
    assert float(((after - ref).abs() / ref.abs().amax(-1, keepdim=True).clamp_min(1e-3)).max()) < 2e-5
    with pytest.raises(ValueError):
        BatchedPolicy(2, 5, 30, 8, device=DEV, gemm="fp16x3")                 # fc1 not a multiple of 4
    assert BatchedPolicy(2, 5, 30, 8, device=DEV).gemm == "fp32"
    assert lib.risvec_policy_layer1_split16(4, 2, 5, 30, 16, 16, 16, 16, 16, 16, None) == N.ERR_SHAPE


def test_fused_mlp_edges_and_weight_updates():
    """risvec_policy_mlp: row counts that leave partly empty wavefronts, zero observations (LayerNorm-1
    variance from the bias row alone), a weight update picked up through the tensors' version counters,
    20 heads (V = 16), and the driver's 512 / 256 sizes at E = 32 768 against the three-launch form."""
    from ris_vec_marl_amd import BatchedPolicy
    for (V, F1, F2, E) in ((8, 512, 256, 1), (8, 512, 256, 65), (16, 64, 128, 300), (3, 512, 256, 257)):
        pol = BatchedPolicy(V, 5, F1, F2, device=DEV, seed=E)
        assert pol.gemm == "fused"
        with torch.no_grad():
            pol.Wh.mul_(50.0); pol.ln1_w.uniform_(0.5, 1.5); pol.ln1_b.uniform_(-0.2, 0.2); pol.ln2_w.uniform_(0.5, 1.5)
        obs = torch.rand(E, V, 5, device=DEV) * 1.2
        obs[0] = 0.0
        a, b = pol.forward_heads(obs), pol.forward_heads_torch(obs)
        scale = b.abs().amax(dim=-1, keepdim=True).clamp_min(1e-3)
        assert float(((a - b).abs() / scale).max()) < 2e-5, (V, F1, F2, E)
        with torch.no_grad():
            pol.W1.mul_(0.5); pol.b1.add_(0.1); pol.W2.mul_(1.25)
        a2, b2 = pol.forward_heads(obs), pol.forward_heads_torch(obs)
        assert not torch.equal(a, a2)
        assert float(((a2 - b2).abs() / b2.abs().amax(dim=-1, keepdim=True).clamp_min(1e-3)).max()) < 2e-5
    # nearly constant fc1 (tiny weights): the scaling of the fc1 operand is capped so that the scaled LayerNorm-1
    # output stays inside float16
    pol = BatchedPolicy(4, 5, 512, 256, device=DEV, seed=3)
    with torch.no_grad():
        pol.W1.mul_(1e-3); pol.b1.mul_(1e-3); pol.Wh.mul_(50.0); pol.ln1_w.fill_(3.0); pol.ln1_b.fill_(-0.5)
    obs = torch.rand(300, 4, 5, device=DEV)
    a, b = pol.forward_heads(obs), pol.forward_heads_torch(obs)
    assert torch.isfinite(a).all()
    assert float(((a - b).abs() / b.abs().amax(dim=-1, keepdim=True).clamp_min(1e-3)).max()) < 5e-5    # rstd ~ 300 amplifies input rounding
    E, V = 32768, 8
    pol = BatchedPolicy(V, 5, 512, 256, device=DEV, seed=1)
    ref = BatchedPolicy(V, 5, 512, 256, device=DEV, seed=1, gemm="fp32")
    with torch.no_grad():
        pol.Wh.mul_(50.0); ref.Wh.mul_(50.0)
    obs = torch.rand(E, V, 5, device=DEV) * 1.2
    a, b = pol.forward_heads(obs), ref.forward_heads(obs)
    assert float(((a - b).abs() / b.abs().amax(dim=-1, keepdim=True).clamp_min(1e-3)).max()) < 2e-5
