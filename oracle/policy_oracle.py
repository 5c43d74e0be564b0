"""CPU ORACLE for the batched `choose_action` -- TEST INFRASTRUCTURE ONLY.

float64 NumPy restatement of `Simulation-MARL-BCD/sac_agent.py` (SAC below): `PolicyNetwork.forward`
(SAC:62-78), `sample_normal` without reparameterisation (SAC:80-131) and `Agent.choose_action`
(SAC:187-225), for one agent's network applied to a batch of observations.  The random draws are
explicit inputs: `eps` ~ N(0,1) [B, 2] (what `Normal.sample` draws, SAC:85) and `expo` ~ Exp(1) [B, N]
(what `F.gumbel_softmax` draws before `-log`, SAC:110-113).

    Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may
    import it.  The product package must not (and does not).

Parity status: PINNED by `tests/golden/policy_*.npz` (tools/capture_golden_policy.py builds the
reference's own PolicyNetwork objects, calls their `sample_normal` under a fixed torch seed and
checks that the recorded draws reproduce the reference's outputs bit for bit before saving).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np

LN_EPS = 1e-5                      # nn.LayerNorm default (SAC:33-34)
FP32_MIN_HALF = float(np.finfo(np.float32).min) / 2.0     # SAC:103


def layer_norm(x: np.ndarray, w: np.ndarray, b: np.ndarray) -> np.ndarray:
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)          # biased, as torch
    return (x - mu) / np.sqrt(var + LN_EPS) * w + b


def forward(wts: Dict[str, np.ndarray], state: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """SAC:62-78.  wts: torch state_dict names -> arrays (Linear weights are [out, in])."""
    W = {k: np.asarray(v, dtype=np.float64) for k, v in wts.items()}
    x = np.asarray(state, dtype=np.float64)
    x = np.maximum(layer_norm(x @ W["fc1.weight"].T + W["fc1.bias"], W["bn1.weight"], W["bn1.bias"]), 0.0)
    x = np.maximum(layer_norm(x @ W["fc2.weight"].T + W["fc2.bias"], W["bn2.weight"], W["bn2.bias"]), 0.0)
    mu = x @ W["mu.weight"].T + W["mu.bias"]
    log_std = np.clip(x @ W["log_std.weight"].T + W["log_std.bias"], -20.0, 2.0)
    logits = x @ W["intent_logits.weight"].T + W["intent_logits.bias"]
    return mu, log_std, logits


def mask_logits(logits: np.ndarray, mask: Optional[np.ndarray]) -> np.ndarray:
    """SAC:91-104: blocked entries -> finfo(float32).min / 2; an all-zero mask row is opened up."""
    if mask is None:
        return logits
    m = np.array(np.broadcast_to(np.asarray(mask, dtype=np.float64), logits.shape))
    m[m.sum(-1) == 0] = 1.0
    return np.where(m <= 0, FP32_MIN_HALF, logits)


def choose_action(wts: Dict[str, np.ndarray], state: np.ndarray, mask: Optional[np.ndarray], tau: float,
                  eps: np.ndarray, expo: np.ndarray, hard: bool = False) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """SAC:187-225 -> (power_action [B,2], intent_probs [B,N], intent_onehot [B,N]).  `hard`: the
    straight-through form `y_hard - y_soft + y_soft` of F.gumbel_softmax(hard=True) (SAC:110-113 with
    `gumbel_hard`, switched on by the driver at TRAIN:1816-1818), evaluated in float32 like the reference
    because its value IS the float32 rounding of that expression."""
    mu, log_std, logits = forward(wts, state)
    x_t = mu + np.exp(log_std) * np.asarray(eps, dtype=np.float64)        # Normal(mu, std).sample()
    power = np.tanh(x_t)
    ml = mask_logits(logits, mask)
    g = (ml + -np.log(np.asarray(expo, dtype=np.float64))) / float(tau)  # gumbel_softmax, soft (SAC:110-113)
    g = g - g.max(-1, keepdims=True)
    y = np.exp(g)
    y = y / y.sum(-1, keepdims=True)
    onehot = np.zeros_like(y)
    onehot[np.arange(len(y)), y.argmax(-1)] = 1.0
    if hard:
        y32 = y.astype(np.float32)
        y = ((onehot.astype(np.float32) - y32) + y32).astype(np.float64)
    return power, y, onehot


def top2_gap(y: np.ndarray) -> np.ndarray:
    s = np.sort(y, axis=-1)
    return s[..., -1] - s[..., -2] if y.shape[-1] > 1 else np.ones(y.shape[:-1])
