"""CPU restatement (numpy) of the KIVI attention cache state machine -- TEST INFRASTRUCTURE ONLY.

Follows cuda_supported_gear/modeling_llama_kivi.py:81-289 from the point where q / k / v (post-RoPE, fp16) are known up to the
tensor handed to o_proj: K fp16 window quantized as a block of `residual` tokens (:149-162), V sliding window whose oldest token
is quantized per token once it holds residual + 1 (:200-213), prompt split (:222-248).  Building blocks: oracle.py
(quant_pack_lastdim / gemv_outer, pinned by the golden fixtures F1 / F7).  The state machine as a whole is pinned by
tests/golden/f8_ref_kivi_*.npz: traces made by executing the reference's own LlamaAttention_KIVI.forward source step by step
(tests/golden/make_f8_ref.py), which tests/test_oracle_golden.py holds this restatement to."""
import math

import numpy as np

from . import oracle as orc
from .attention_oracle import add16, mm16, rep


class KiviAttentionOracle:
    def __init__(self, n_heads, n_kv_heads, head_dim, group, bits, residual):
        self.H, self.Hkv, self.D, self.g, self.b, self.R = n_heads, n_kv_heads, head_dim, group, bits, residual
        self.n_rep = n_heads // n_kv_heads
        self.c = None

    def _q(self, x):
        r = orc.quant_pack_lastdim(x, self.g, self.b, mode=0)
        return r["code"], r["scale"], r["mn"]

    def _softmax(self, w, mask=None):
        w = (w.astype(np.float32) / np.float32(math.sqrt(self.D))).astype(np.float16)
        if mask is not None:
            w = np.maximum(add16(w, mask), np.float16(np.finfo(np.float16).min))
        w32 = w.astype(np.float32)
        w32 = np.exp(w32 - w32.max(-1, keepdims=True))
        return (w32 / w32.sum(-1, keepdims=True)).astype(np.float16)

    def prefill(self, q, k, v, mask=None):
        R, T = self.R, k.shape[2]
        a = self._softmax(mm16(q, rep(k, self.n_rep).transpose(0, 1, 3, 2)), mask)
        out = mm16(a, rep(v, self.n_rep))
        c = dict(kc=None, ks=None, km=None, kfull=None, vc=None, vs=None, vm=None, vfull=None, n=T)
        nq = T - T % R
        if nq:
            c["kc"], c["ks"], c["km"] = self._q(np.ascontiguousarray(k[:, :, :nq].transpose(0, 1, 3, 2)))
        c["kfull"] = k[:, :, nq:] if nq < T else None
        if T <= R:
            c["vfull"] = v
        else:
            c["vfull"] = v[:, :, -R:]
            c["vc"], c["vs"], c["vm"] = self._q(np.ascontiguousarray(v[:, :, :-R]))
        self.c = c
        return out

    def decode(self, q, k, v):
        c, R = self.c, self.R
        parts = []
        if c["kc"] is not None:
            parts.append(orc.gemv_outer(q, c["kc"], c["ks"], c["km"], self.g, self.b))
        c["kfull"] = k if c["kfull"] is None else np.concatenate([c["kfull"], k], 2)
        parts.append(mm16(q, rep(c["kfull"], self.n_rep).transpose(0, 1, 3, 2)))
        w = np.concatenate(parts, -1)
        if c["kfull"].shape[2] == R:
            kc, ks, km = self._q(np.ascontiguousarray(c["kfull"].transpose(0, 1, 3, 2)))
            c["kfull"] = None
            if c["kc"] is None:
                c["kc"], c["ks"], c["km"] = kc, ks, km
            else:
                c["kc"], c["ks"], c["km"] = (np.concatenate([c["kc"], kc], 3), np.concatenate([c["ks"], ks], 3),
                                             np.concatenate([c["km"], km], 3))
        a = self._softmax(w)
        c["vfull"] = np.concatenate([c["vfull"], v], 2)
        nfull = c["vfull"].shape[2]
        if c["vc"] is None:
            out = mm16(a, rep(c["vfull"], self.n_rep))
        else:
            out = orc.gemv_outer(np.ascontiguousarray(a[..., :-nfull]), c["vc"], c["vs"], c["vm"], self.g, self.b)
            out = add16(out, mm16(a[..., -nfull:], rep(c["vfull"], self.n_rep)))
        if nfull > R:
            vc, vs, vm = self._q(np.ascontiguousarray(c["vfull"][:, :, :1]))
            c["vfull"] = c["vfull"][:, :, 1:]
            if c["vc"] is None:
                c["vc"], c["vs"], c["vm"] = vc, vs, vm
            else:
                c["vc"], c["vs"], c["vm"] = (np.concatenate([c["vc"], vc], 2), np.concatenate([c["vs"], vs], 2),
                                             np.concatenate([c["vm"], vm], 2))
        c["n"] += 1
        return out
