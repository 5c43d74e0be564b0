#!/usr/bin/env python3
"""Capture golden vectors for the batched choose_action from the REFERENCE.
Runs ONLY in the CPU build container.

Imports the reference's own `sac_agent.PolicyNetwork` (CPU torch), builds one network per agent as
`Agent.__init__` does (sac_agent.py:165-167; small hidden sizes keep the fixtures small -- the
architecture is size-agnostic), and calls `sample_normal(state, reparameterize=False, mask)` exactly as
`choose_action` does (sac_agent.py:210) under a fixed torch seed.  The draws it consumed are then
re-drawn from the same seed (`normal_` [B,2], `exponential_` [B,N], the calls Normal.sample and
F.gumbel_softmax make) and VERIFIED to reproduce the reference's outputs bit for bit before anything is
saved.  Fixtures hold weights, inputs, draws and outputs only.
"""
from __future__ import annotations

import os
import sys
import tempfile

import numpy as np

REF_DIR = "/root/reference/Simulation-MARL-BCD"
OUT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
if not os.path.isfile(os.path.join(REF_DIR, "sac_agent.py")):
    sys.exit("capture_golden_policy: reference not present (this tool only runs in the build container)")
sys.dont_write_bytecode = True
sys.path.insert(0, REF_DIR)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import sac_agent as REF  # noqa: E402  (the reference itself)


def capture(tag, V, fc1, fc2, B, seed):
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    tmp = tempfile.mkdtemp(prefix="risvec_policy_")          # the constructor creates its checkpoint dir
    rec = dict(state=[], mask=[], has_mask=[], tau=[], eps=[], expo=[], power=[], probs=[], onehot=[], mu=[],
               log_std=[], logits=[], hard=[])
    weights = {}
    for a in range(V):
        net = REF.PolicyNetwork(3e-4, 5, fc1, fc2, 2, V, name="policy", agent_label=a, chkpt_dir=tmp)
        net = net.to("cpu")
        net.device = torch.device("cpu")
        with torch.no_grad():                                  # move the net away from its near-zero heads
            for p in (net.mu.weight, net.log_std.weight, net.intent_logits.weight):
                p.uniform_(-0.4, 0.4)
            net.bn1.weight.uniform_(0.5, 1.5); net.bn1.bias.uniform_(-0.2, 0.2)
            net.bn2.weight.uniform_(0.5, 1.5); net.bn2.bias.uniform_(-0.2, 0.2)
        tau = float(rng.choice([2.0, 1.0, 0.5]))
        net.tau.fill_(tau)
        state = torch.from_numpy(rng.uniform(0, 1.2, (B, 5)).astype(np.float32))
        has_mask = a % 3 != 2
        hard = a % 4 == 3                                       # straight-through one-hot (driver: tau <= 0.3 and gumbel_hard)
        net.gumbel_hard = hard
        mask = torch.from_numpy((rng.uniform(size=(B, V)) < 0.6).astype(np.float32))
        mask[0] = 0.0                                           # an all-zero row: the reference opens it up
        net.eval()
        s = seed * 100 + a
        with torch.no_grad():
            torch.manual_seed(s)
            power, y, _, _, _ = net.sample_normal(state, reparameterize=False, mask=mask if has_mask else None)
            mu, log_std, logits = net.forward(state)
            # re-draw what sample_normal consumed and verify
            torch.manual_seed(s)
            eps = torch.empty(B, 2).normal_()
            expo = torch.empty(B, V).exponential_()
            x_t = eps * log_std.exp() + mu
            ml = logits
            if has_mask:
                m = mask.clone()
                m[m.sum(-1) == 0] = 1.0
                ml = logits.masked_fill(m <= 0, torch.finfo(logits.dtype).min / 2)
            y2 = ((ml + -expo.log()) / tau).softmax(-1)
            if hard:                                            # F.gumbel_softmax(hard=True): y_hard - y_soft + y_soft
                y_hard = torch.zeros_like(y2).scatter_(-1, y2.max(-1, keepdim=True)[1], 1.0)
                y2 = y_hard - y2 + y2
            assert torch.equal(torch.tanh(x_t), power), "normal draws not reproduced"
            assert torch.equal(y2, y), "gumbel draws not reproduced"
            onehot = F.one_hot(torch.argmax(y, -1), num_classes=V).float()      # choose_action, :215-216
        for k, v in net.state_dict().items():
            if k != "tau":
                weights["a%d.%s" % (a, k)] = v.numpy().copy()
        for k, v in dict(state=state, mask=mask, has_mask=has_mask, hard=hard, tau=tau, eps=eps, expo=expo, power=power, probs=y,
                         onehot=onehot, mu=mu, log_std=log_std, logits=logits).items():
            rec[k].append(v.numpy() if hasattr(v, "numpy") else v)
    np.savez_compressed(os.path.join(OUT_DIR, "policy_%s.npz" % tag), V=V, fc1=fc1, fc2=fc2, B=B,
                        **{k: np.asarray(v) for k, v in rec.items()}, **weights)
    print("policy_%s: %d agents, batch %d, hidden %d/%d" % (tag, V, B, fc1, fc2))


if __name__ == "__main__":
    os.makedirs(OUT_DIR, exist_ok=True)
    capture("8", 8, 48, 32, 96, 7)
    capture("4", 4, 40, 24, 33, 8)
