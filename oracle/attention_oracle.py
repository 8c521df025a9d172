"""CPU restatement (numpy) of the GEAR attention cache state machine -- TEST INFRASTRUCTURE ONLY.

Follows cuda_supported_gear/modeling_llamagear.py:177-484 (LlamaAttention_GEAR.forward) from the point where
q / k / v (post-RoPE, fp16) are known up to the tensor handed to o_proj, with the build's documented stances on
reference defects B1 (all code columns packed) and B2 (K factors approximate the true error).  The reference's own
forward cannot be imported under the installed transformers, so this restatement is pinned
only through its building blocks (oracle.py functions, each checked against golden vectors) -- "parity unpinned"
at the level of the whole state machine.
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
