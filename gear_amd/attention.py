"""Decode attention straight over compressed payloads (gear_amd.compress.Payload) -- the fused
"decompress-into-attention" kernel of gear_amd/csrc/attention.hip behind a small Python function."""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import _lib as L
from .compress import Payload


def decode_attention(q: torch.Tensor, pk: Optional[Payload], pv: Optional[Payload], k_window: Optional[torch.Tensor] = None,
                     v_window: Optional[torch.Tensor] = None, return_lse: bool = False):
    """q fp16 [B,Hq,1,128]; pk / pv the K / V payloads of the compressed tokens ([B,Hkv,T,128]); k_window / v_window
    fp16 [B,Hkv,W,128] the most recent W <= 64 uncompressed tokens.  Returns fp16 [B,Hq,1,128] =
    softmax(q Khat^T / sqrt(128)) Vhat over compressed + window tokens (fp32 softmax), optionally with the LSE."""
    assert q.dim() == 4 and q.shape[2] == 1 and q.dtype == torch.float16
    B, Hq, _, D = q.shape
    q = q.contiguous()
    L.require_gpu(q)
    lib = L.load()
    W = 0 if k_window is None else k_window.shape[2]
    if W:
        k_window, v_window = k_window.contiguous(), v_window.contiguous()
        L.require_gpu(k_window, v_window)
    if pk is not None:
        assert pk.kind == "k" and pv is not None and pv.kind == "v" and pk.shape == pv.shape
        assert pk.bits == pv.bits and pk.group == pv.group and pk.mode == pv.mode
        _, Hkv, T, _ = pk.shape
        fpi = 32 // pk.bits
        args = dict(kcode=pk.code, kscale=pk.scale, kmn=pk.mn, kP=pk.P, kQ=pk.Q, koidx=pk.oidx, koval=pk.oval,
                    vcode=pv.code, vscale=pv.scale, vmn=pv.mn, vP=pv.P, vQ=pv.Q, voidx=pv.oidx, voval=pv.oval)
        ldk, lsk = T // fpi, T // pk.group
        bits, group, mode = pk.bits, pk.group, pk.mode
        rk, rv, kk, kv = pk.rank, pv.rank, pk.k_out, pv.k_out
        # chunk index of the outlier lists (contexts the 128-token-chunk kernel handles)
        args["kochunk"] = pk.chunk_index() if T <= 8192 else None
        args["vochunk"] = pv.chunk_index() if T <= 8192 else None
    else:
        Hkv, T = k_window.shape[1], 0
        args = {n: None for n in ("kcode kscale kmn kP kQ koidx koval vcode vscale vmn vP vQ voidx voval kochunk vochunk").split()}
        ldk = lsk = 0
        bits, group, mode, rk, rv, kk, kv = 2, 64, 0, 0, 0, 0, 0
    out = torch.empty((B, Hq, 1, D), dtype=torch.float16, device=q.device)
    lse = torch.empty((B, Hq), dtype=torch.float32, device=q.device) if return_lse else None
    wsb = lib.gear_attn_decode_workspace(B, Hq, T, bits)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=q.device)
    p = L.ptr
    rc = lib.gear_attn_decode_idx(p(q), p(args["kcode"]), p(args["kscale"]), p(args["kmn"]), p(args["kP"]), p(args["kQ"]),
                                  p(args["koidx"]), p(args["koval"]), p(args["kochunk"]), p(args["vcode"]), p(args["vscale"]),
                                  p(args["vmn"]), p(args["vP"]), p(args["vQ"]), p(args["voidx"]), p(args["voval"]),
                                  p(args["vochunk"]), p(k_window), p(v_window),
                                  B, Hq, Hkv, D, T, W, ldk, lsk, T, T, T, group, bits, mode, rk, rv, kk, kv,
                                  1.0 / math.sqrt(D), p(out), p(lse), p(ws), wsb, L.stream_ptr())
    L.check(rc, "gear_attn_decode")
    return (out, lse) if return_lse else out


def decode_attention_f16(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, T: Optional[int] = None, return_lse: bool = False):
    """The uncompressed baseline (the reference harness's model "None", cuda_supported_gear/test.py:41-62): q fp16 [B,Hq,1,128] over
    an fp16 cache k, v [B,Hkv,tcap,128] whose first T tokens are valid (default: all).  Same split / merge kernels and the same
    grouped-query mapping as decode_attention, so the two times are comparable."""
    assert q.dim() == 4 and q.shape[2] == 1 and q.dtype == torch.float16 and k.dtype == torch.float16 and k.shape == v.shape
    B, Hq, _, D = q.shape
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    L.require_gpu(q, k, v)
    Hkv, tcap = k.shape[1], k.shape[2]
    T = tcap if T is None else T
    lib = L.load()
    out = torch.empty((B, Hq, 1, D), dtype=torch.float16, device=q.device)
    lse = torch.empty((B, Hq), dtype=torch.float32, device=q.device) if return_lse else None
    wsb = lib.gear_attn_decode_workspace(B, Hq, T, 2)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=q.device)
    rc = lib.gear_attn_decode_f16(L.ptr(q), L.ptr(k), L.ptr(v), B, Hq, Hkv, D, T, tcap, 1.0 / math.sqrt(D), L.ptr(out), L.ptr(lse),
                                  L.ptr(ws), wsb, L.stream_ptr(q))
    L.check(rc, "gear_attn_decode_f16")
    return (out, lse) if return_lse else out
