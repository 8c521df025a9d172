"""CPU restatement (numpy) of the GEAR attention cache state machine -- TEST INFRASTRUCTURE ONLY.

Follows cuda_supported_gear/modeling_llamagear.py:177-484 (LlamaAttention_GEAR.forward) from the point where
q / k / v (post-RoPE, fp16) are known up to the tensor handed to o_proj, with the build's documented stances on
reference defects B1 (all code columns packed) and B2 (K factors approximate the true error).  The reference's module
cannot be imported whole under the installed transformers; the pin is tests/golden/f8_ref_*.npz:
traces made by executing the reference's OWN forward / key_compression / value_compression / matmul_withlrap source
(extracted with ast by tests/golden/make_f8_ref.py, model plumbing stubbed), which tests/test_oracle_golden.py holds
this restatement to -- the KIVI-method cases with nothing of the reference replaced, the low-rank cases with the two
defective glue functions replaced by their documented stance.
"""
import math

import numpy as np

from . import oracle as orc


def mm16(a, b):
    """torch fp16 matmul: fp32 accumulate, one fp16 rounding."""
    return np.matmul(a.astype(np.float32), b.astype(np.float32)).astype(np.float16)


def add16(a, b):
    return (a.astype(np.float32) + b.astype(np.float32)).astype(np.float16)


def rep(t, n):
    return t if n == 1 else np.repeat(t, n, axis=-3)


def matmul_withlrap(group_size, a, code, scale, mn, bits, pbase, qbase, type="key", n_rep=1):
    """modeling_llamagear.py:54-111: dequant GEMV + low-rank correction, fp16 matmuls.  pbase / qbase in the reference's
    format: [None] | [prefill] | [prefill, stacked [nbuf, B, H, ., r]].  a [B,Hq,1,K]."""
    r1 = orc.gemv_outer(np.ascontiguousarray(a), code, scale, mn, group_size, bits)
    if pbase[0] is None:
        return r1
    if type == "key":          # r1[.., :Tp] += (a Q0) P0^T ; r1[.., Tp + 64 i ..] += (a Q1[i]) P1[i]^T   (:64-85)
        parts = [mm16(mm16(a, rep(qbase[0], n_rep)), rep(pbase[0], n_rep).transpose(0, 1, 3, 2))]
        if len(pbase) > 1:
            for P, Qf in zip(pbase[1], qbase[1]):
                parts.append(mm16(mm16(a, rep(Qf, n_rep)), rep(P, n_rep).transpose(0, 1, 3, 2)))
        return add16(r1, np.concatenate(parts, axis=-1))
    # value: out += (a[:, :Tp] Q0) P0^T + sum_i (a_blk_i Q1[i]) P1[i]^T   (:87-108)
    tp = qbase[0].shape[-2]
    out = add16(r1, mm16(mm16(a[..., :tp], rep(qbase[0], n_rep)), rep(pbase[0], n_rep).transpose(0, 1, 3, 2)))
    if len(pbase) > 1:
        bl = qbase[1].shape[-2]
        acc = None
        for i, (P, Qf) in enumerate(zip(pbase[1], qbase[1])):
            t = mm16(mm16(a[..., tp + i * bl:tp + (i + 1) * bl], rep(Qf, n_rep)), rep(P, n_rep).transpose(0, 1, 3, 2))
            acc = t if acc is None else add16(acc, t)       # torch: result4.sum(dim=0) of fp16 terms
        out = add16(out, acc)
    return out


class GearAttentionOracle:
    def __init__(self, n_heads, n_kv_heads, head_dim, cfg, draw_p0):
        """cfg: the compress_config dict (compress_method, group_size, residual, quantize_bit, rank, rankv, loop).
        draw_p0(B, H, S, Dm, r) -> float32 [B,H,Dm,r]: must mirror the product's RNG use (torch.rand P then Q)."""
        self.H, self.Hkv, self.D, self.cfg, self.draw = n_heads, n_kv_heads, head_dim, cfg, draw_p0
        self.n_rep = n_heads // n_kv_heads
        self.lowrank = "gearl" in cfg["compress_method"] or "gearsl" in cfg["compress_method"]
        self.c = None

    # modeling_llamagear.py:23-37 / :39-53
    def _kcomp(self, kt):
        g, b = self.cfg["group_size"], self.cfg["quantize_bit"]
        if not self.lowrank:
            return orc.key_compression(kt, g, b, 0, 0, None, lowrank_on=False)
        B, H, D, T = kt.shape
        return orc.key_compression(kt, g, b, self.cfg["rank"], self.cfg["loop"], self.draw(B, H, D, T, self.cfg["rank"]))

    def _vcomp(self, v):
        g, b = self.cfg["group_size"], self.cfg["quantize_bit"]
        if not self.lowrank:
            return orc.value_compression(v, g, b, 0, 0, None, lowrank_on=False)
        B, H, T, D = v.shape
        return orc.value_compression(v, g, b, self.cfg["rankv"], self.cfg["loop"], self.draw(B, H, T, D, self.cfg["rankv"]))

    def prefill(self, q, k, v, mask=None):
        """q [B,H,T,D], k/v [B,Hkv,T,D] fp16 -> attention output [B,H,T,D] fp16 (:386-456)."""
        R = self.cfg["residual"]
        T = k.shape[2]
        w = mm16(q, rep(k, self.n_rep).transpose(0, 1, 3, 2))
        w = (w.astype(np.float32) / np.float32(math.sqrt(self.D))).astype(np.float16)
        if mask is not None:
            w = np.maximum(add16(w, mask), np.float16(np.finfo(np.float16).min))
        w32 = w.astype(np.float32)
        w32 = np.exp(w32 - w32.max(-1, keepdims=True))
        a = (w32 / w32.sum(-1, keepdims=True)).astype(np.float16)
        out = mm16(a, rep(v, self.n_rep))
        c = dict(kc=None, ks=None, km=None, kp=None, kq=None, kfull=None, vc=None, vs=None, vm=None, vp=None, vq=None,
                 vfull=None, n=T)
        nq = T - T % R
        if T >= R:
            c["kc"], c["ks"], c["km"], p, qf = self._kcomp(np.ascontiguousarray(k[:, :, :nq].transpose(0, 1, 3, 2)))
            c["kp"], c["kq"] = [p], [qf]
            c["kfull"] = k[:, :, nq:] if nq < T else None
        else:
            c["kfull"] = k
        if T > R:
            c["vc"], c["vs"], c["vm"], p, qf = self._vcomp(np.ascontiguousarray(v[:, :, :nq]))
            c["vp"], c["vq"] = [p], [qf]
            c["vfull"] = v[:, :, nq:] if nq < T else None
        else:
            c["vfull"] = v
        self.c = c
        return out

    def _lr_key(self, q, kp, kq):
        """(a Q) P^T per factor pair, fp16 matmuls (:64-85)."""
        outs = []
        for P, Qf in zip(kp, kq):
            if P is None:
                return None
            r2 = mm16(q, rep(Qf, self.n_rep))
            outs.append(mm16(r2, rep(P, self.n_rep).transpose(0, 1, 3, 2)))
        return np.concatenate(outs, axis=-1)

    def decode(self, q, k, v):
        """one token: q [B,H,1,D], k/v [B,Hkv,1,D] -> [B,H,1,D] (:209-378)."""
        c, R, g, b = self.c, self.cfg["residual"], self.cfg["group_size"], self.cfg["quantize_bit"]
        parts = []
        if c["kc"] is not None:
            s = orc.gemv_outer(q, c["kc"], c["ks"], c["km"], g, b)
            lr = self._lr_key(q, c["kp"], c["kq"])
            if lr is not None:
                s = add16(s, lr)
            parts.append(s)
        c["kfull"] = k if c["kfull"] is None else np.concatenate([c["kfull"], k], axis=2)
        parts.append(mm16(q, rep(c["kfull"], self.n_rep).transpose(0, 1, 3, 2)))
        w = np.concatenate(parts, axis=-1)
        w = (w.astype(np.float32) / np.float32(math.sqrt(self.D))).astype(np.float16)
        if c["kfull"].shape[2] == R:
            kc, ks, km, p, qf = self._kcomp(np.ascontiguousarray(c["kfull"].transpose(0, 1, 3, 2)))
            c["kfull"] = None
            if c["kc"] is None:
                c["kc"], c["ks"], c["km"], c["kp"], c["kq"] = kc, ks, km, [p], [qf]
            else:
                c["kc"] = np.concatenate([c["kc"], kc], 3)
                c["ks"] = np.concatenate([c["ks"], ks], 3)
                c["km"] = np.concatenate([c["km"], km], 3)
                c["kp"].append(p)
                c["kq"].append(qf)
        w32 = w.astype(np.float32)
        w32 = np.exp(w32 - w32.max(-1, keepdims=True))
        a = (w32 / w32.sum(-1, keepdims=True)).astype(np.float16)
        c["vfull"] = v if c["vfull"] is None else np.concatenate([c["vfull"], v], axis=2)
        nfull = c["vfull"].shape[2]
        if c["vc"] is None:
            out = mm16(a, rep(c["vfull"], self.n_rep))
        else:
            aq = np.ascontiguousarray(a[..., :-nfull])
            out = orc.gemv_outer(aq, c["vc"], c["vs"], c["vm"], g, b)
            t0 = 0
            for P, Qf in zip(c["vp"], c["vq"]):
                if P is None:
                    break
                tl = Qf.shape[2]
                r2 = mm16(aq[..., t0:t0 + tl], rep(Qf, self.n_rep))
                out = add16(out, mm16(r2, rep(P, self.n_rep).transpose(0, 1, 3, 2)))
                t0 += tl
            out = add16(out, mm16(a[..., -nfull:], rep(c["vfull"], self.n_rep)))
        if nfull == R:
            vc, vs, vm, p, qf = self._vcomp(np.ascontiguousarray(c["vfull"]))
            c["vfull"] = None
            if c["vc"] is None:
                c["vc"], c["vs"], c["vm"], c["vp"], c["vq"] = vc, vs, vm, [p], [qf]
            else:
                c["vc"] = np.concatenate([c["vc"], vc], 2)
                c["vs"] = np.concatenate([c["vs"], vs], 2)
                c["vm"] = np.concatenate([c["vm"], vm], 2)
                c["vp"].append(p)
                c["vq"].append(qf)
        c["n"] += 1
        return out
