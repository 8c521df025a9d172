"""Drop-in counterpart of cuda_supported_gear/quant/matmul.py: fp16 activation x packed-int K/V batched GEMV.

`cuda_bmm_fA_qB_outer` keeps its reference name (callers import it by name, modeling_llamagear.py:10); it runs the
gfx950 kernel in gear_amd/csrc/gemv.hip through the C ABI.  Unlike the reference it does NOT re-lay-out the whole
packed cache on every call (matmul.py:205, :215-216: transpose(1,2).contiguous() of qB / scales / zeros).
"""
from __future__ import annotations

import torch

from .. import _lib as L


def cuda_bmm_fA_qB_outer(group_size: int, fA: torch.Tensor, qB: torch.Tensor, scales: torch.Tensor,
                         zeros: torch.Tensor, bits: int, mqa: bool = False) -> torch.Tensor:
    """C = fA x dequant(qB)   (matmul.py:178-220).

    fA [B,nh,M,K] fp16 (M must be 1 -- the reference kernel ignores blockIdx.z, defect B4);
    qB [B,nkv,K,N/fpi] int32; scales / zeros [B,nkv,K,N/g] (fp16, or float32 for payloads produced in "fp32" mode).
    nkv == nh (MHA), nkv == 1 with mqa=True (the reference's MQA path), or any divisor of nh (GQA).
    Returns fp16 [B,nh,M,N]."""
    assert len(fA.shape) == 4 and len(qB.shape) == 4        # matmul.py:199
    assert bits in [2, 4]                                   # matmul.py:217
    B, nh, M, K = fA.shape
    if M != 1:
        raise L.GearError("cuda_bmm_fA_qB_outer supports q_len == 1 only (decode GEMV)")
    if fA.dtype != torch.float16:
        raise L.GearError(f"fA must be float16 (got {fA.dtype})")
    feat_per_int = 32 // bits
    N = qB.shape[-1] * feat_per_int
    qB = qB.reshape(B, -1, K, qB.shape[-1])
    nkv = qB.shape[1]
    if mqa:
        assert nkv == 1
    assert nh % nkv == 0
    scales = scales.reshape(B, nkv, K, -1)
    zeros = zeros.reshape(B, nkv, K, -1)
    mode = 0 if scales.dtype == torch.float16 else 1
    fA, qB, scales, zeros = fA.contiguous(), qB.contiguous(), scales.contiguous(), zeros.contiguous()
    L.require_gpu(fA, qB, scales, zeros)
    lib = L.load()
    BA = B * nh
    ws_bytes = lib.gear_gemv_outer_workspace(BA, K, N, bits)
    ws = torch.empty((max(ws_bytes, 4) // 4,), dtype=torch.float32, device=fA.device)
    c = torch.empty((B, nh, M, N), dtype=torch.float16, device=fA.device)
    rc = lib.gear_gemv_outer(L.ptr(fA), L.ptr(qB), L.ptr(scales), L.ptr(zeros), BA, nh // nkv, K, N, group_size, bits,
                             mode, 0, 0, L.ptr(c), L.ptr(ws), ws_bytes, L.stream_ptr())
    L.check(rc, "gear_gemv_outer")
    return c


def gemv_forward_cuda_outer_dim(in_feats: torch.Tensor, kernel: torch.Tensor, scaling_factors: torch.Tensor, zeros: torch.Tensor,
                                bit: int, group_size: int, nh: int, mqa: bool) -> torch.Tensor:
    """kivi_gemv.gemv_forward_cuda_outer_dim (csrc/pybind.cpp:5-8, gemv_cuda.h:13-21) on ITS argument layout: in_feats [BS, M = 1, K]
    fp16, kernel [BS', N / fpi, K] int32, scaling_factors / zeros [BS', N / group, K] -- BS' = BS, or BS / nh with mqa -- i.e. what
    matmul.py:205, :215-216 produce by transpose(1, 2).contiguous().  Returns [BS, 1, N] fp16.  For a caller that keeps the
    reference's cuda_bmm_fA_qB_outer line by line; the operator above needs no re-layout."""
    assert in_feats.dim() == 3 and kernel.dim() == 3 and bit in (2, 4)
    BS, M, K = in_feats.shape
    if M != 1:
        raise L.GearError("gemv_forward_cuda_outer_dim supports M == 1 only (decode GEMV; the reference kernel ignores blockIdx.z)")
    N = kernel.shape[1] * (32 // bit)
    n_rep = nh if mqa else BS // kernel.shape[0]
    mode = 0 if scaling_factors.dtype == torch.float16 else 1
    in_feats, kernel, scaling_factors, zeros = in_feats.contiguous(), kernel.contiguous(), scaling_factors.contiguous(), zeros.contiguous()
    L.require_gpu(in_feats, kernel, scaling_factors, zeros)
    out = torch.empty((BS, 1, N), dtype=torch.float16, device=in_feats.device)
    rc = L.load().gear_gemv_outer_dim(L.ptr(in_feats), L.ptr(kernel), L.ptr(scaling_factors), L.ptr(zeros), BS, n_rep, K, N, group_size,
                                     bit, mode, L.ptr(out), L.stream_ptr())
    L.check(rc, "gear_gemv_outer_dim")
    return out


def triton_bmm_fA_qB_outer(group_size: int, fA, qB, scales, zeros, bits: int) -> torch.Tensor:
    """matmul.py:112-175 (the Triton alternative of the same operator): same kernel here."""
    return cuda_bmm_fA_qB_outer(group_size, fA, qB, scales, zeros, bits)
