#!/usr/bin/env python3
"""Capture golden vectors for the replay buffer + action marshalling from the
REFERENCE.  Runs ONLY in the CPU build container.

  * `Simulation-MARL-BCD/buffer.py` is imported as is and driven with random transitions past the
    wrap-around; the indices its `np.random.choice` draws are recorded.
  * the marshalling statements of `marl_train_bcd.py`'s step loop (the script cannot be imported)
    are extracted by AST and executed in memory on recorded policy outputs: the
    `intent_probs_mat` stack + zero diagonal, the `action_for_env` map, and the `action_to_store`
    concatenation.

Only inputs and outputs are written (tests/golden/replay_*.npz).
"""
from __future__ import annotations

import ast
import os
import sys
import types

import numpy as np

REF_DIR = "/root/reference/Simulation-MARL-BCD"
REF = os.path.join(REF_DIR, "marl_train_bcd.py")
OUT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
if not os.path.isfile(REF):
    sys.exit("capture_golden_replay: reference not present (this tool only runs in the build container)")
sys.dont_write_bytecode = True
sys.path.insert(0, REF_DIR)
import buffer as REFBUF  # noqa: E402  (the reference itself)

TREE = ast.parse(open(REF, encoding="utf-8").read(), filename=REF)


def _stores(stmt):
    return {n.id for n in ast.walk(stmt) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Store)}


def step_loop_body():
    loops = [n for n in ast.walk(TREE) if isinstance(n, ast.For) and isinstance(n.target, ast.Name)
             and n.target.id == "i_step"]
    return next(l for l in loops if any("action_to_store" in _stores(s) for s in l.body)).body


def pick(body, name, extra_after=0):
    k = next(k for k, s in enumerate(body) if isinstance(s, ast.Assign) and name in _stores(s))
    return body[k:k + 1 + extra_after]


def marshal_code():
    body = step_loop_body()
    stmts = []
    stmts += pick(body, "intent_probs_mat", extra_after=1)      # stack + fill_diagonal
    assert isinstance(stmts[-1], ast.Expr) and "fill_diagonal" in ast.dump(stmts[-1])
    stmts += pick(body, "action_for_env", extra_after=1)        # zeros + the mapping loop
    assert isinstance(stmts[-1], ast.For)
    for name in ("power_np", "probs_np", "action_cat", "action_to_store"):
        stmts += pick(body, name)
    return compile(ast.Module(body=stmts, type_ignores=[]), REF, "exec")


def capture_marshal(tag, V, seed, n=64):
    code = marshal_code()
    rng = np.random.default_rng(seed)
    rec = dict(power=[], probs=[], floor=[], action_env=[], store=[])
    for c in range(n):
        power = rng.uniform(-1.3, 1.3, (V, 2)).astype(np.float32)
        if c % 5 == 0:
            power[rng.integers(0, V)] = [1.0, -1.0]
        probs = rng.dirichlet(np.ones(V), V).astype(np.float32)
        floor = float(rng.choice([0.10, 0.02, 0.0, 0.97, -0.2]))
        ns = dict(np=np, config=types.SimpleNamespace(n_veh=V), env=types.SimpleNamespace(cpu_share_floor=floor),
                  intent_probs_list=[probs[i].copy() for i in range(V)],
                  marl_power_actions=[power[i].copy() for i in range(V)])
        exec(code, ns)
        rec["power"].append(power); rec["probs"].append(probs); rec["floor"].append(floor)
        rec["action_env"].append(np.asarray(ns["action_for_env"], dtype=np.float64))
        rec["store"].append(np.asarray(ns["action_to_store"], dtype=np.float32))
    np.savez_compressed(os.path.join(OUT_DIR, "replay_marshal_%s.npz" % tag), V=V,
                        **{k: np.asarray(v) for k, v in rec.items()})
    print("replay_marshal_%s: %d cases" % (tag, n))


def capture_buffer(tag, V, cap, n_store, n_sample, seed):
    rng = np.random.default_rng(seed)
    n_in, n_act = 5, V + 2
    buf = REFBUF.ReplayBuffer(cap, n_in, n_act, V)
    st = rng.normal(size=(n_store, n_in * V)).astype(np.float32)
    ac = rng.normal(size=(n_store, n_act * V)).astype(np.float32)
    rg = rng.normal(size=n_store).astype(np.float32)
    rl = rng.normal(size=(n_store, V)).astype(np.float32)
    st2 = rng.normal(size=(n_store, n_in * V)).astype(np.float32)
    dn = rng.uniform(size=n_store) < 0.1
    mk = (rng.uniform(size=(n_store, V * V)) < 0.7).astype(np.float32)
    for i in range(n_store):
        buf.store_transition(st[i], ac[i], float(rg[i]), rl[i], st2[i], bool(dn[i]), mk[i])
    drawn = []
    orig = np.random.choice

    def choice(*a, **k):
        v = orig(*a, **k)
        drawn.append(np.asarray(v))
        return v
    np.random.seed(seed)
    np.random.choice = choice
    try:
        out = buf.sample_buffer(n_sample)
    finally:
        np.random.choice = orig
    np.savez_compressed(
        os.path.join(OUT_DIR, "replay_buffer_%s.npz" % tag), V=V, cap=cap, state=st, action=ac, reward_g=rg, reward_l=rl,
        state_=st2, done=dn, mask=mk, mem_cntr=buf.mem_cntr, state_memory=buf.state_memory,
        action_memory=buf.action_memory, reward_global_memory=buf.reward_global_memory,
        reward_local_memory=buf.reward_local_memory, new_state_memory=buf.new_state_memory,
        terminal_memory=buf.terminal_memory, mask_memory=buf.mask_memory, batch=drawn[0],
        s_states=out[0], s_actions=out[1], s_rewards_g=out[2], s_rewards_l=out[3], s_states_=out[4], s_dones=out[5],
        s_masks=out[6])
    print("replay_buffer_%s: %d stores into %d slots, %d sampled" % (tag, n_store, cap, n_sample))


if __name__ == "__main__":
    os.makedirs(OUT_DIR, exist_ok=True)
    capture_marshal("8", 8, 31)
    capture_marshal("4", 4, 32, 32)
    capture_buffer("8_wrap", 8, 100, 257, 64, 41)       # wraps 2.57 times
    capture_buffer("8_partial", 8, 500, 123, 32, 42)    # not yet full
    capture_buffer("4_exact", 4, 64, 64, 16, 43)        # exactly full
