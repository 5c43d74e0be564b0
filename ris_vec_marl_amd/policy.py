"""f3: batched `choose_action` -- the reference's per-agent SAC policies
(`Simulation-MARL-BCD/sac_agent.py`, SAC below: `PolicyNetwork` SAC:9-131, `Agent.choose_action`
SAC:187-225) evaluated for all E envs and all V agents at once, on the GPU.

The driver calls `agents[i].choose_action(obs_i, mask=mask_row)` V times per step, each a batch-1
forward with a host round trip (TRAIN:1373-1384).  Here the V networks' weights are stacked and the
three small GEMMs of `forward` (5 -> fc1 -> fc2 -> {mu, log_std, intent_logits}) run as `torch.bmm`
(rocBLAS: plain library GEMMs); everything after them -- clamp, Normal sample + tanh, logits mask,
Gumbel-softmax, arg-max one-hot AND the marshalling of TRAIN:1386-1396, 1601-1608, 1776-1784 -- is one
hand-written HIP launch (`risvec_policy_sample`, csrc/k_policy.hip).  No CPU path.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Mapping, Optional, Sequence, Tuple

import torch

from . import _native as N

_LAYERS = ("fc1", "fc2", "mu", "log_std", "intent_logits")


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class BatchedPolicy:
    """V stacked `PolicyNetwork`s (SAC:9-60): input_dims -> fc1 -> LayerNorm -> ReLU -> fc2 -> LayerNorm
    -> ReLU -> heads mu[n_actions], log_std[n_actions], intent_logits[n_agents]."""

    GEMM_MODES = ("fused", "fp16x3", "fp32")

    def __init__(self, n_agents: int, input_dims: int = 5, fc1_dims: int = 512, fc2_dims: int = 256,
                 n_actions: int = 2, device="cuda", seed: int = 0, env_offset: int = 0, gemm: Optional[str] = None):
        """gemm: how the forward runs.  "fp32": hand-written fc1 launch, float32 library GEMM for fc1 x fc2,
        hand-written heads launch.  "fp16x3": the same three launches with both GEMM operands split into
        float16 high and low parts and the three significant partial products taken in ONE float16 GEMM
        with float32 accumulation (K three times as long) -- float32-GEMM accuracy (2^-22 per product) at
        the fp16 matrix-core rate; see `risvec_policy_layer1_split16`.  "fused" (default where the kernel
        is built for the shape: input_dims <= 5, fc1 % 32 == 0, fc2 in {128, 256}, 4 + n_agents <= 24):
        the whole forward in one hand-written MFMA kernel with the same split product, the hidden layers
        never leaving the chip (`risvec_policy_mlp`).  Default: the first of fused / fp16x3 / fp32 that
        supports the shape."""
        N.load()
        if n_actions != 2:
            raise ValueError("the reference's power head has 2 outputs (offload, local); got %d" % n_actions)
        self.device = N.resolve_device(device)
        if self.device.type != "cuda" or not torch.cuda.is_available():
            raise RuntimeError("ris_vec_marl_amd needs a HIP device; there is no CPU fallback")
        self.n_agents, self.input_dims, self.fc1_dims, self.fc2_dims = int(n_agents), int(input_dims), int(fc1_dims), int(fc2_dims)
        self.seed, self.env_offset, self._calls = int(seed), int(env_offset), 0
        split_ok = self.fc1_dims % 4 == 0 and self.fc1_dims <= 1024 and self.input_dims <= 8
        fused_ok = bool(N.load().risvec_policy_mlp_supported(self.input_dims, self.fc1_dims, self.fc2_dims, 4 + self.n_agents))
        self.gemm = gemm if gemm is not None else ("fused" if fused_ok else ("fp16x3" if split_ok else "fp32"))
        if (self.gemm not in self.GEMM_MODES or (self.gemm == "fp16x3" and not split_ok)
                or (self.gemm == "fused" and not fused_ok)):
            raise ValueError("gemm=%r is not available for input_dims=%d fc1=%d fc2=%d n_agents=%d (fp16x3: fc1 %% 4 == 0 <= "
                             "1024, input_dims <= 8; fused: input_dims <= 5, fc1 %% 32 == 0, fc2 in {128, 256}, n_agents <= 20)"
                             % (gemm, self.input_dims, self.fc1_dims, self.fc2_dims, self.n_agents))
        self._w2_split = (None, None)                         # (key, [V, 3 fc1, fc2] float16)
        self._fused_w = (None, None)                          # (key, (Wc, G, W2f))
        V, dev = self.n_agents, self.device
        g = torch.Generator(device="cpu").manual_seed(seed)

        def uni(*shape, r):
            return ((torch.rand(*shape, generator=g) * 2 - 1) * r).to(dev)
        f1, f2, f3 = 1.0 / math.sqrt(fc1_dims), 1.0 / math.sqrt(fc2_dims), 0.003        # SAC:36-50
        H = 4 + V
        self.W1, self.b1 = uni(V, input_dims, fc1_dims, r=f1), uni(V, 1, fc1_dims, r=f1)
        self.W2, self.b2 = uni(V, fc1_dims, fc2_dims, r=f2), uni(V, 1, fc2_dims, r=f2)
        self.Wh, self.bh = uni(V, fc2_dims, H, r=f3), uni(V, 1, H, r=f3)               # [mu | log_std | intent_logits]
        self.ln1_w, self.ln1_b = torch.ones(V, 1, fc1_dims, device=dev), torch.zeros(V, 1, fc1_dims, device=dev)
        self.ln2_w, self.ln2_b = torch.ones(V, 1, fc2_dims, device=dev), torch.zeros(V, 1, fc2_dims, device=dev)
        self.tau = torch.full((V,), 2.0, device=dev)                                    # SAC:54 (annealed by the driver)
        self.gumbel_hard = torch.zeros(V, dtype=torch.uint8, device=dev)                # SAC:56, per agent

    # ------------------------------------------------------------------ weights
    def load_agent_state_dict(self, agent: int, sd: Mapping[str, object]) -> None:
        """Take agent `agent`'s weights from a reference `PolicyNetwork.state_dict()` (tensors or arrays;
        Linear weights are [out, in] there).  Load checkpoints with `torch.load(..., weights_only=True)`."""
        t = lambda k: torch.as_tensor(sd[k], dtype=torch.float32).to(self.device)   # noqa: E731
        a = int(agent)
        self.W1[a], self.b1[a, 0] = t("fc1.weight").T, t("fc1.bias")
        self.W2[a], self.b2[a, 0] = t("fc2.weight").T, t("fc2.bias")
        self.Wh[a] = torch.cat([t("mu.weight"), t("log_std.weight"), t("intent_logits.weight")], 0).T
        self.bh[a, 0] = torch.cat([t("mu.bias"), t("log_std.bias"), t("intent_logits.bias")], 0)
        self.ln1_w[a, 0], self.ln1_b[a, 0] = t("bn1.weight"), t("bn1.bias")
        self.ln2_w[a, 0], self.ln2_b[a, 0] = t("bn2.weight"), t("bn2.bias")
        if "tau" in sd:
            self.tau[a] = float(torch.as_tensor(sd["tau"]))

    def set_tau(self, tau: float, gumbel_hard: bool = False) -> None:
        """The driver's temperature annealing (TRAIN:1804-1815 sets every agent's `policy.tau`) and its
        switch to the straight-through one-hot form (TRAIN:1816-1818: `want_hard and cur_tau <= 0.3`;
        pass the resulting boolean)."""
        self.tau.fill_(float(tau))
        self.gumbel_hard.fill_(1 if gumbel_hard else 0)

    def _split_w2(self) -> torch.Tensor:
        """[ hi(W2) ; (W2 - hi(W2)) 2^5 ; hi(W2) 2^-6 ] along K as float16, rebuilt when W2 changes."""
        key = (self.W2.data_ptr(), self.W2._version)
        if self._w2_split[0] != key:
            hi = self.W2.to(torch.float16)
            lo = (self.W2 - hi.float()) * 32.0
            self._w2_split = (key, torch.cat([hi, lo.to(torch.float16), (hi.float() * 0.015625).to(torch.float16)], 1).contiguous())
        return self._w2_split[1]

    @staticmethod
    def _split_scaled(w: torch.Tensor, target: float, max_shift: Optional[torch.Tensor] = None):
        """(hi, lo, 2^-s): w 2^s with its largest entry in [target, 2 target) per agent, split into float16
        hi + lo (the scaling keeps lo in the float16 normal range; powers of two cancel exactly)."""
        amax = w.abs().amax(dim=tuple(range(1, w.dim()))).clamp_min(1e-30)
        shift = torch.floor(torch.log2(target / amax)).clamp(-14, 14)
        if max_shift is not None:
            shift = torch.minimum(shift, max_shift)
        ws = w.float() * torch.exp2(shift).reshape(-1, *([1] * (w.dim() - 1)))
        hi = ws.to(torch.float16)
        return hi, (ws - hi.float()).to(torch.float16), torch.exp2(-shift).float()

    def _fused_weights(self):
        """(G, W1F, W2f, unscale, WhF, wh_unscale) of `risvec_policy_mlp` (layouts: include/risvec.h), rebuilt
        when a weight changes: the Gram matrix of the centred fc1 rows (LayerNorm-1 statistics in closed
        form, float64 here), and the three layers' weights scaled, split into float16 hi + lo and laid out in
        MFMA fragment order."""
        key = tuple((t.data_ptr(), t._version) for t in (self.W1, self.b1, self.ln1_w, self.ln1_b, self.W2, self.Wh))
        if self._fused_w[0] != key:
            V, IN, F1, F2, dev = self.n_agents, self.input_dims, self.fc1_dims, self.fc2_dims, self.device
            wb = torch.cat([self.W1, self.b1], 1).double()                          # [V, IN+1, F1]
            wc = torch.zeros(V, 6, F1, dtype=torch.float64, device=dev)
            wc[:, :IN + 1] = wb - wb.mean(-1, keepdim=True)
            gram = torch.bmm(wc, wc.transpose(1, 2)) / F1
            # fc1 + LayerNorm-1 operand [V, F1, 16]
            op1 = torch.zeros(V, F1, 16, dtype=torch.float64, device=dev)
            op1[:, :, :IN + 1] = (wc[:, :IN + 1] * self.ln1_w.double()).transpose(1, 2)
            op1[:, :, IN + 1] = self.ln1_b.double()[:, 0]
            # the scaled LayerNorm-1 output (<= sqrt(F1) |ln1_w| + |ln1_b|) must stay inside float16
            bound = math.sqrt(F1) * self.ln1_w.abs().amax(dim=(1, 2)) + self.ln1_b.abs().amax(dim=(1, 2))
            h1, l1, u1 = self._split_scaled(op1.float(), 16.0, torch.floor(torch.log2(3.0e4 / bound.clamp_min(1e-30))))
            # (v, t, g, r, h, j) -> (v, g, t, h, r, j): lane = 32 h + r
            w1f = torch.stack([h1, l1], 1).reshape(V, 2, F1 // 32, 32, 2, 8).permute(0, 2, 1, 4, 3, 5).reshape(V, F1 // 32, 2, 64, 8)
            h2, l2, u2 = self._split_scaled(self.W2, 64.0)
            # hidden feature f = 32 g + 16 u + 8 jh + 4 h + jl: (v, t, g, u, jh, h, jl, m, r) -> (v, g, u, t, m, h, r, jh, jl)
            frag = torch.stack([h2, l2], 1).reshape(V, 2, F1 // 32, 2, 2, 2, 4, F2 // 32, 32) \
                .permute(0, 2, 3, 1, 7, 5, 8, 4, 6).contiguous()
            whp = torch.zeros(V, F2, 32, device=dev)
            whp[:, :, :4 + V] = self.Wh
            hh, hl, uh = self._split_scaled(whp, 64.0)
            # feature f = 32 m + 16 u + 8 jh + 4 h + jl: (v, t, m, u, jh, h, jl, r) -> (v, m, u, t, h, r, jh, jl)
            hfrag = torch.stack([hh, hl], 1).reshape(V, 2, F2 // 32, 2, 2, 2, 4, 32).permute(0, 2, 3, 1, 5, 7, 4, 6).contiguous()
            # the weight stream, per group of 32 hidden features: fc1 operand of the NEXT group (8 KiB slot), two chunks
            MT, NG = F2 // 32, F1 // 32
            stream = torch.zeros(V, NG, 8 + 4 * MT, 64, 8, dtype=torch.float16, device=dev)
            stream[:, :-1, 0:2] = w1f[:, 1:]
            stream[:, :, 8:] = frag.reshape(V, NG, 4 * MT, 64, 8)
            self._fused_w = (key, (gram.float().contiguous(), w1f[:, 0].contiguous(), stream, (u1 * u2).contiguous(), hfrag,
                                   uh.contiguous()))
        return self._fused_w[1]

    # ------------------------------------------------------------------ forward
    def forward_heads(self, obs: torch.Tensor) -> torch.Tensor:
        """SAC:62-78 for every agent: obs [E,V,input_dims] -> [V,E,4+V] rows (mu, log_std (unclamped),
        intent_logits).  fc1 + LayerNorm + ReLU is one hand-written launch (its "GEMM" has K = 5), the
        fc1 x fc2 product is a batched rocBLAS GEMM, and LayerNorm + ReLU + the three heads are a second
        hand-written launch, so the normalised hidden layers never make an extra HBM round trip."""
        E, V = int(obs.shape[0]), self.n_agents
        x = obs.to(self.device, torch.float32).contiguous()
        lib, stream = N.load(), torch.cuda.current_stream(self.device).cuda_stream
        if self.gemm == "fused":
            gram, w1f, frag, unscale, hfrag, hunscale = self._fused_weights()
            heads = torch.empty(V, E, 4 + V, device=self.device)
            N.check(lib.risvec_policy_mlp(E, V, self.input_dims, self.fc1_dims, self.fc2_dims, 4 + V, x.data_ptr(),
                                          gram.data_ptr(), w1f.data_ptr(), frag.data_ptr(), unscale.data_ptr(),
                                          self.b2.data_ptr(), self.ln2_w.data_ptr(), self.ln2_b.data_ptr(), hfrag.data_ptr(),
                                          hunscale.data_ptr(), self.bh.data_ptr(), heads.data_ptr(), stream))
            return heads
        if self.gemm == "fp16x3":
            w2s = self._split_w2()
            h1 = torch.empty(V, E, 3 * self.fc1_dims, dtype=torch.float16, device=self.device)
            N.check(lib.risvec_policy_layer1_split16(E, V, self.input_dims, self.fc1_dims, x.data_ptr(), self.W1.data_ptr(),
                                                     self.b1.data_ptr(), self.ln1_w.data_ptr(), self.ln1_b.data_ptr(),
                                                     h1.data_ptr(), stream))
            g2 = torch.bmm(h1, w2s, out_dtype=torch.float32)     # [V,E,fc2]: fp16 MFMA GEMM, float32 accumulate + output
        else:
            h1 = torch.empty(V, E, self.fc1_dims, device=self.device)
            N.check(lib.risvec_policy_layer1(E, V, self.input_dims, self.fc1_dims, x.data_ptr(), self.W1.data_ptr(),
                                             self.b1.data_ptr(), self.ln1_w.data_ptr(), self.ln1_b.data_ptr(), h1.data_ptr(),
                                             stream))
            g2 = torch.bmm(h1, self.W2)                          # [V,E,fc2]: library GEMM (its bias is added downstream)
        heads = torch.empty(V, E, 4 + V, device=self.device)
        N.check(lib.risvec_policy_heads(E, V, self.fc2_dims, 4 + V, g2.data_ptr(), self.b2.data_ptr(), self.ln2_w.data_ptr(),
                                        self.ln2_b.data_ptr(), self.Wh.data_ptr(), self.bh.data_ptr(), heads.data_ptr(),
                                        stream))
        return heads

    def forward_heads_torch(self, obs: torch.Tensor) -> torch.Tensor:
        """The same forward with library kernels only (torch.bmm / layer_norm): the fp32 reference the
        hand-written launches are tested against."""
        x = obs.to(self.device, torch.float32).transpose(0, 1)                          # [V,E,in]
        F = torch.nn.functional
        h = torch.baddbmm(self.b1, x, self.W1)
        h = torch.relu(F.layer_norm(h, (self.fc1_dims,)) * self.ln1_w + self.ln1_b)
        h = torch.baddbmm(self.b2, h, self.W2)
        h = torch.relu(F.layer_norm(h, (self.fc2_dims,)) * self.ln2_w + self.ln2_b)
        return torch.baddbmm(self.bh, h, self.Wh).contiguous()

    def choose_action(self, obs: torch.Tensor, mask: Optional[torch.Tensor] = None, eps: Optional[torch.Tensor] = None,
                      expo: Optional[torch.Tensor] = None, cpu_share_floor: Optional[float] = None,
                      want_onehot: bool = True, out: Optional[Sequence[torch.Tensor]] = None) -> Tuple[torch.Tensor, ...]:
        """`agents[i].choose_action(obs_i, mask=mask[i])` for every env and agent (SAC:187-225).
        obs [E,V,5]; mask [E,V,V] uint8/bool (the NOMA mask: row v = who agent v may pick) or None.
        Returns (power_action [E,V,2], intent_probs [E,V,V], intent_onehot [E,V,V] or None); with
        `cpu_share_floor` given, additionally the marshalled (action_env [E,2,V], p_off01 [E,V],
        action_store [E,V*(V+2)]) exactly as `marshal_actions` would produce them -- same launch.
        `eps` [E,V,2] / `expo` [E,V,V] inject the N(0,1) / Exp(1) draws (parity); default Philox.
        `out` = (action_env, p_off01, action_store): write the marshalled outputs into these
        caller-owned contiguous float32 tensors (the ones bound to the env step / NOMA grouping /
        replay store launchers) instead of new ones."""
        E, V = int(obs.shape[0]), self.n_agents
        if tuple(obs.shape) != (E, V, self.input_dims):
            raise ValueError("obs must have shape [E, %d, %d]" % (V, self.input_dims))
        heads = self.forward_heads(obs)
        dev = self.device
        mk = None
        if mask is not None:
            mk = mask.to(dev)
            mk = (mk != 0).to(torch.uint8) if mk.dtype != torch.uint8 else mk
            mk = mk.reshape(E, V, V).contiguous()
        ep = None if eps is None else eps.to(dev, torch.float32).reshape(E, V, 2).contiguous()
        ex = None if expo is None else expo.to(dev, torch.float32).reshape(E, V, V).contiguous()
        power = torch.empty(E, V, 2, device=dev)
        probs = torch.empty(E, V, V, device=dev)
        onehot = torch.empty(E, V, V, device=dev) if want_onehot else None
        marsh = cpu_share_floor is not None
        if out is not None:
            if not marsh:
                raise ValueError("choose_action: out= needs cpu_share_floor (the marshalled outputs)")
            a_env, p01, a_store = out
            for t, shape in ((a_env, (E, 2, V)), (p01, (E, V)), (a_store, (E, V * (V + 2)))):
                if t.dtype != torch.float32 or t.device != dev or not t.is_contiguous() or tuple(t.shape) != shape:
                    raise ValueError("choose_action: out tensors must be contiguous float32 [E,2,V], [E,V], [E,V*(V+2)] on %s" % dev)
        else:
            a_env = torch.empty(E, 2, V, device=dev) if marsh else None
            p01 = torch.empty(E, V, device=dev) if marsh else None
            a_store = torch.empty(E, V * (V + 2), device=dev) if marsh else None
        self._calls += 1
        N.check(N.load().risvec_policy_sample(
            E, V, self.env_offset, heads.data_ptr(), _ptr(mk), self.tau.data_ptr(), self.gumbel_hard.data_ptr(), _ptr(ep),
            _ptr(ex), self.seed,
            self._calls, float(cpu_share_floor) if marsh else 0.0, power.data_ptr(), probs.data_ptr(), _ptr(onehot),
            _ptr(a_env), _ptr(p01), _ptr(a_store), torch.cuda.current_stream(dev).cuda_stream))
        out = (power, probs, onehot)
        return out + (a_env, p01, a_store) if marsh else out

    # ------------------------------------------------------------------ checkpoint
    _KEYS = ("W1", "b1", "W2", "b2", "Wh", "bh", "ln1_w", "ln1_b", "ln2_w", "ln2_b", "tau", "gumbel_hard")

    def state_dict(self) -> dict:
        return {k: getattr(self, k).detach().cpu().clone() for k in self._KEYS}

    def load_state_dict(self, sd: Mapping[str, torch.Tensor]) -> None:
        for k in self._KEYS:
            getattr(self, k).copy_(sd[k].to(self.device))
