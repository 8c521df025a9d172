"""Drop-in counterpart of cuda_supported_gear/quant/new_pack.py on hand-written gfx950 kernels.

Same function names, argument order, return shapes and assert behaviour as the reference; the Triton / eager
torch implementation is replaced by libgear_hip.so (include/gear_hip.h).  Function names keep the reference's
`triton_` prefix because callers import them by name (cuda_supported_gear/modeling_llamagear.py:9).

Extra keyword `mode` (not in the reference): "fp16" (default) = the reference's fp16-stepwise arithmetic, packed
payload bit-exact; "fp32" = the simulated path's fp32 arithmetic with float32 scale/mn.
"""
from __future__ import annotations

import torch

from .. import _lib as L

_MODES = {"fp16": L.MODE_FP16_STEPWISE, "fp32": L.MODE_FP32, 0: 0, 1: 1}


def _sdt(mode):
    return torch.float16 if _MODES[mode] == 0 else torch.float32


def _check_half(t, name):
    if t.dtype != torch.float16:
        raise L.GearError(f"{name} must be float16 (got {t.dtype})")  # reference: c10::Error from data_ptr<at::Half>()


# ------------------------------------------------------------------------------------------------ a1 / a2
def _quant_lastdim(data, group_size, bit, mode, with_error):
    assert len(data.shape) == 4                      # new_pack.py:218 / :254
    B, nh, D, T = data.shape
    assert T % group_size == 0                       # new_pack.py:222 / :258
    _check_half(data, "data")
    data = data.contiguous()
    L.require_gpu(data)
    m = _MODES[mode]
    fpi = 32 // bit
    code = torch.empty((B, nh, D, T // fpi), dtype=torch.int32, device=data.device)
    scale = torch.empty((B, nh, D, T // group_size), dtype=_sdt(mode), device=data.device)
    mn = torch.empty_like(scale)
    err = torch.empty((B * nh * D, T // group_size, group_size), dtype=torch.float16, device=data.device) \
        if with_error else None
    rc = L.load().gear_quant_pack_lastdim(L.ptr(data), B * nh * D, T, group_size, bit, m, L.ptr(code), L.ptr(scale),
                                          L.ptr(mn), L.ptr(err), L.stream_ptr())
    L.check(rc, "gear_quant_pack_lastdim")
    return code, scale, mn, err


def triton_quantize_and_pack_along_last_dim(data: torch.Tensor, group_size: int, bit: int, mode="fp16"):
    """new_pack.py:217-250.  data fp16 [B,nh,D,T] -> (code i32 [B,nh,D,T/fpi], scale [B,nh,D,T/g], mn [B,nh,D,T/g])."""
    code, scale, mn, _ = _quant_lastdim(data, group_size, bit, mode, False)
    return code, scale, mn


def triton_quantize_and_pack_along_last_dim_witherror(data: torch.Tensor, group_size: int, bit: int, mode="fp16"):
    """new_pack.py:253-288 -> (code, scale, mn, error fp16 [B*nh*D, T/g, g]).

    Divergence from the reference (defect B1): ALL packed columns are written; the reference's
    pack grid uses num_groups instead of T (:283) and leaves most of `code` zero."""
    return _quant_lastdim(data, group_size, bit, mode, True)


# ------------------------------------------------------------------------------------------------ a3
def quant_and_pack_vcache(v: torch.Tensor, group_size: int, bits: int, mode="fp16"):
    """new_pack.py:30-48.  v [B,nh,T,D] -> (code [B,nh,T,D/fpi], scale [B,nh,T,D/g,1], mn [B,nh,T,D/g,1])."""
    shape = v.shape
    assert len(shape) == 4
    assert v.shape[-1] % group_size == 0
    code, scale, mn, _ = _quant_lastdim(v, group_size, bits, mode, False)
    return code, scale.unsqueeze(-1), mn.unsqueeze(-1)


def quant_and_pack_kcache(k: torch.Tensor, group_size: int, bits: int, mode="fp16"):
    """new_pack.py:8-27.  k [B,nh,T,D] -> (code [B,nh,T/fpi,D], scale [B,nh,T/g,1,D], mn [B,nh,T/g,1,D])."""
    assert len(k.shape) == 4
    B, nh, T, D = k.shape
    assert T % group_size == 0
    _check_half(k, "k")
    k = k.contiguous()
    L.require_gpu(k)
    fpi = 32 // bits
    code = torch.empty((B, nh, T // fpi, D), dtype=torch.int32, device=k.device)
    scale = torch.empty((B, nh, T // group_size, 1, D), dtype=_sdt(mode), device=k.device)
    mn = torch.empty_like(scale)
    rc = L.load().gear_quant_pack_k(L.ptr(k), B * nh, T, D, group_size, bits, _MODES[mode], L.ptr(code), L.ptr(scale),
                                    L.ptr(mn), None, L.stream_ptr())
    L.check(rc, "gear_quant_pack_k")
    return code, scale, mn


def unpack_and_dequant_vcache(v_code, scale, mn, group_size: int, bits: int, mode="fp16"):
    """new_pack.py:69-83 -> fp16 [B,nh,T,D]."""
    assert bits in [2, 4, 8]
    assert len(v_code.shape) == 4
    v_code, scale, mn = v_code.contiguous(), scale.contiguous(), mn.contiguous()
    L.require_gpu(v_code, scale, mn)
    B, nh, T, nw = v_code.shape
    Dm = nw * (32 // bits)
    out = torch.empty((B, nh, T, Dm), dtype=torch.float16, device=v_code.device)
    rc = L.load().gear_unpack_dequant_lastdim(L.ptr(v_code), L.ptr(scale), L.ptr(mn), B * nh * T, Dm, group_size, bits,
                                              _MODES[mode], L.ptr(out), L.stream_ptr())
    L.check(rc, "gear_unpack_dequant_lastdim")
    return out


def unpack_and_dequant_kcache(k_code, scale, mn, group_size: int, bits: int, mode="fp16"):
    """new_pack.py:51-66 -> fp16 [B,nh,T,D]."""
    assert bits in [2, 4, 8]
    assert len(k_code.shape) == 4
    k_code, scale, mn = k_code.contiguous(), scale.contiguous(), mn.contiguous()
    L.require_gpu(k_code, scale, mn)
    B, nh, nw, D = k_code.shape
    T = nw * (32 // bits)
    out = torch.empty((B, nh, T, D), dtype=torch.float16, device=k_code.device)
    rc = L.load().gear_unpack_dequant_k(L.ptr(k_code), L.ptr(scale), L.ptr(mn), B * nh, T, D, group_size, bits,
                                        _MODES[mode], L.ptr(out), L.stream_ptr())
    L.check(rc, "gear_unpack_dequant_k")
    return out


def pack_tensor(data: torch.Tensor, bits: int, pack_dim: int):
    """new_pack.py:86-107: OR-pack 32/bits int codes along pack_dim, element j at bits [bits*(j%fpi), ...).
    Format helper (not on the hot path): vectorised torch integer ops on whatever device `data` lives on."""
    shape = data.shape
    feat_per_int = 32 // bits
    assert bits in [2, 4, 8], "Only 2, 4, 8 bits are supported"
    assert shape[pack_dim] % feat_per_int == 0, "Dimension length must be divisible by number of features per int"
    d = data.to(torch.int64).movedim(pack_dim, -1)
    d = d.reshape(d.shape[:-1] + (d.shape[-1] // feat_per_int, feat_per_int))
    shifts = torch.arange(feat_per_int, device=data.device, dtype=torch.int64) * bits
    w = ((d << shifts).sum(dim=-1)) & 0xFFFFFFFF
    w = torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32)
    return w.movedim(-1, pack_dim).contiguous()


def unpack_tensor(v_code: torch.Tensor, bits: int, pack_dim: int):
    """new_pack.py:110-129 -> int16 codes."""
    assert bits in [2, 4, 8]
    feat_per_int = 32 // bits
    c = v_code.movedim(pack_dim, -1)
    shifts = torch.arange(feat_per_int, device=v_code.device, dtype=torch.int32) * bits
    u = (c.unsqueeze(-1) >> shifts) & (0xFF >> (8 - bits))
    u = u.reshape(c.shape[:-1] + (c.shape[-1] * feat_per_int,)).to(torch.int16)
    return u.movedim(-1, pack_dim).contiguous()


# ------------------------------------------------------------------------------------------------ a4
def headwise_lrap(tensor: torch.Tensor, rank, loop, p_base: torch.Tensor = None):
    """new_pack.py:291-311.  tensor [B,nh,S,Dm] -> (p_base [B,nh,Dm,rank], q_base [B,nh,S,rank]) in tensor.dtype;
    the approximation is q_base @ p_base^T.

    The initial basis is drawn exactly like the reference (torch.rand on the CPU generator, then a discarded q_base
    draw, :296-297) unless `p_base` (float32 [B,nh,Dm,rank]) is supplied.  Bases are per (batch, head): the
    reference's `p_base[0]` indexing (:301, :304) broadcasts batch 0's basis and is only correct for B == 1
    (defect B3) -- identical results for B == 1."""
    from .. import compress
    dtype = tensor.dtype
    batch, num_head, seq_len, head_dim = tensor.shape
    if p_base is None:
        p_base = compress.draw_p0(batch, num_head, seq_len, head_dim, rank, tensor.device)
    out_dtype = torch.float16 if dtype == torch.float16 else torch.float32
    P, Q = compress.lowrank(tensor, rank, loop, p_base, transposed=False, out_dtype=out_dtype)
    return P.type(dtype), Q.type(dtype)
