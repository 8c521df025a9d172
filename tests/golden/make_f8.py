"""Fixture F8: a 130-step single-layer decode trace of the attention cache state machine
(prefill 200 tokens -> crosses two 64-token blocks), produced by oracle/attention_oracle.py -- the restatement of
cuda_supported_gear/modeling_llamagear.py:177-484 with the documented stances on defects B1 / B2.  This trace is NOT reference
output: it is a regression pin of the ORACLE on shapes the reference rejects (GQA) -- a later edit of the restatement that
changes its behaviour fails tests/test_oracle_golden.py::test_f8_*.  The pin against the reference's own executed forward is
tests/golden/make_f8_ref.py -> f8_ref_*.npz.
Inputs (q, k, v per step, the bases P0) are stored, not re-drawn.   python tests/golden/make_f8.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.attention_oracle import GearAttentionOracle  # noqa: E402

CASES = {"gearl_b2_mha": ("gearlKIVI", 2, 2, 2), "gearl_b4_gqa": ("gearlKIVI", 4, 4, 2), "kivi_b2_mha": ("KIVI", 2, 2, 2)}
D, TP, STEPS, RANK = 128, 200, 130, 4


def run(case, store=None, load=None):
    method, bits, H, Hkv = CASES[case]
    cc = dict(compress_method=method, group_size=64, residual=64, quantize_bit=bits, rank=RANK, rankv=RANK, loop=3)
    rng = np.random.default_rng(1234)
    p0s = []

    def draw(B, Hh, S, Dm, r):
        if load is not None:
            p = load[f"{case}_P0_{len(p0s)}"]
        else:
            p = rng.random((B, Hh, Dm, r), dtype=np.float32)
        p0s.append(p)
        return p

    o = GearAttentionOracle(H, Hkv, D, cc, draw)
    if load is not None:
        q, k, v = load[f"{case}_q"], load[f"{case}_k"], load[f"{case}_v"]
    else:
        q = (rng.standard_normal((1, H, TP + STEPS, D)) * 0.5).astype(np.float16)
        k = (rng.standard_normal((1, Hkv, TP + STEPS, D)) * 0.5).astype(np.float16)
        v = (rng.standard_normal((1, Hkv, TP + STEPS, D)) * 0.5).astype(np.float16)
    mask = np.triu(np.full((TP, TP), np.finfo(np.float16).min, np.float16), 1)[None, None]
    outs = [o.prefill(q[:, :, :TP], k[:, :, :TP], v[:, :, :TP], mask)[:, :, -1:]]
    for i in range(STEPS):
        t = TP + i
        outs.append(o.decode(q[:, :, t:t + 1], k[:, :, t:t + 1], v[:, :, t:t + 1]))
    out = np.concatenate(outs, 2)                       # [1, H, 1 + STEPS, D]: last prompt position, then every decode step
    c = o.c
    state = np.array([c["n"], 0 if c["kc"] is None else c["kc"].shape[3], 0 if c["kfull"] is None else c["kfull"].shape[2],
                      0 if c["vc"] is None else c["vc"].shape[2], len(c["kp"] or []), len(c["vp"] or [])])
    if store is not None:
        store.update({f"{case}_q": q, f"{case}_k": k, f"{case}_v": v, f"{case}_out": out, f"{case}_state": state})
        for i, p in enumerate(p0s):
            store[f"{case}_P0_{i}"] = p
    return out, state


if __name__ == "__main__":
    d = {}
    for case in CASES:
        run(case, store=d)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "f8_trace.npz")
    np.savez_compressed(path, **d)
    print(f"f8_trace.npz: {os.path.getsize(path) / 1024:.1f} KiB, {len(d)} arrays")
