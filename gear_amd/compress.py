"""Real (packed) GEAR payloads on the GPU: quantized backbone + low-rank factors + sparse outliers.

This is the build's own composition layer over the C ABI (include/gear_hip.h); the reference only *simulates*
this combination (GenerationBench/.../Simulated/compress_function.py:204-220, :261-333) and its fused path
(cuda_supported_gear/modeling_llamagear.py:23-53) never stores outliers (cache slots 11-12 / 15-16 are None).

Layouts
  V payload : rows = tokens across all heads ("token quantization", compress_function.py:297-333)
      code int32 [B,H,T,D/fpi], scale/mn [B,H,T,D/g], P fp16 [B,H,D,r], Q fp16 [B,H,T,r],
      oidx uint16 [B,T,2k] (index h*D+d inside the token row), oval fp16 [B,T,2k]
  K payload : rows = channels across all tokens ("channel quantization", :261-296), stored channel-major so that
      groups / outlier rows are contiguous and the decode GEMV needs no re-layout (matmul.py:205):
      code int32 [B,H,D,T/fpi], scale/mn [B,H,D,T/g], P fp16 [B,H,D,r], Q fp16 [B,H,T,r],
      oidx uint16 [B,H,D,2k] (token index), oval fp16 [B,H,D,2k]
  For both, the approximation of the per-head [T,D] error is Q @ P^T (the simulated semantics; reference defect B2
  is not reproduced).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib as L

_MODES = {"fp16": 0, "fp32": 1, 0: 0, 1: 1}


@dataclass
class Payload:
    kind: str                 # "k" or "v"
    shape: tuple              # (B, H, T, D) of the fp16 tensor it encodes
    bits: int
    group: int
    mode: int
    code: torch.Tensor
    scale: torch.Tensor
    mn: torch.Tensor
    P: Optional[torch.Tensor] = None
    Q: Optional[torch.Tensor] = None
    oidx: Optional[torch.Tensor] = None
    oval: Optional[torch.Tensor] = None
    k_out: int = 0

    @property
    def rank(self):
        return 0 if self.P is None else self.P.shape[-1]

    def nbytes(self) -> int:
        n = 0
        for t in (self.code, self.scale, self.mn, self.P, self.Q, self.oidx, self.oval):
            if t is not None:
                n += t.numel() * t.element_size()
        return n


def outlier_count(B, H, T, D, sparsity) -> int:
    """k per side per row -- compress_function.py:264-267 / :299-303 (same formula for both layouts, defect B7)."""
    sparsity_num = int(B * H * T * D * sparsity)
    return int(sparsity_num / B / T / 2)


def draw_p0(B, H, S, Dm, rank, device):
    """Initial bases exactly as the reference draws them: torch.rand on the CPU generator, P first, then a Q that is
    discarded (new_pack.py:296-297, compress_function.py:83-84) -- keeps the RNG stream aligned with the reference."""
    p = torch.rand(B, H, Dm, rank)
    _ = torch.rand(B, H, S, rank)
    return p.to(device)


def lowrank(E: torch.Tensor, rank: int, loop: int, P0: torch.Tensor, transposed: bool = False, out_dtype=torch.float16):
    """Power iteration on the GPU.  E [B,H,S,Dm] (or [B,H,Dm,S] if transposed), fp16 / fp32.  Returns P [B,H,Dm,r],
    Q [B,H,S,r]."""
    assert E.dim() == 4
    B, H = E.shape[:2]
    S, Dm = (E.shape[3], E.shape[2]) if transposed else (E.shape[2], E.shape[3])
    E = E.contiguous()
    P0 = P0.to(device=E.device, dtype=torch.float32).contiguous()
    assert tuple(P0.shape) == (B, H, Dm, rank), (tuple(P0.shape), (B, H, Dm, rank))
    L.require_gpu(E, P0)
    if E.dtype not in (torch.float16, torch.float32):
        raise L.GearError(f"lowrank: unsupported dtype {E.dtype}")
    lib = L.load()
    P = torch.empty((B, H, Dm, rank), dtype=out_dtype, device=E.device)
    Q = torch.empty((B, H, S, rank), dtype=out_dtype, device=E.device)
    wsb = lib.gear_lowrank_workspace(B * H, S, Dm, rank)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=E.device)
    rc = lib.gear_lowrank(L.ptr(E), 0 if E.dtype == torch.float16 else 1, 1 if transposed else 0, B * H, S, Dm, rank,
                          loop, L.ptr(P0), L.ptr(P), L.ptr(Q), 0 if out_dtype == torch.float16 else 1, L.ptr(ws), wsb,
                          L.stream_ptr())
    L.check(rc, "gear_lowrank")
    return P, Q


def _compress_rows(x, geom, group, bits, mode, k, want_err):
    n_rows, rows_inner, outer, inner, nseg, seglen, segstride = geom
    fpi = 32 // bits
    dev = x.device
    sdt = torch.float16 if mode == 0 else torch.float32
    code = torch.empty(x.shape[:-1] + (x.shape[-1] // fpi,), dtype=torch.int32, device=dev)
    scale = torch.empty(x.shape[:-1] + (x.shape[-1] // group,), dtype=sdt, device=dev)
    mn = torch.empty_like(scale)
    err = torch.empty_like(x) if want_err else None
    oidx = torch.empty((n_rows, 2 * k), dtype=torch.int16, device=dev) if k > 0 else None
    oval = torch.empty((n_rows, 2 * k), dtype=torch.float16, device=dev) if k > 0 else None
    rc = L.load().gear_compress_rows(L.ptr(x), n_rows, rows_inner, outer, inner, nseg, seglen, segstride, group, bits,
                                     mode, k, L.ptr(code), L.ptr(scale), L.ptr(mn), L.ptr(err), L.ptr(oidx), L.ptr(oval),
                                     None, L.stream_ptr())
    L.check(rc, "gear_compress_rows")
    return code, scale, mn, err, oidx, oval


def compress_value(v: torch.Tensor, bits: int, group: int, k_out: int = 0, rank: int = 0, loop: int = 3,
                   mode="fp32", P0: Optional[torch.Tensor] = None) -> Payload:
    """V [B,H,T,D] fp16 -> Payload (per-token groups along D; outliers per token row across heads)."""
    assert v.dim() == 4 and v.dtype == torch.float16
    v = v.contiguous()
    L.require_gpu(v)
    B, H, T, D = v.shape
    m = _MODES[mode]
    geom = (B * T, T, H * T * D, D, H, D, T * D)
    code, scale, mn, err, oidx, oval = _compress_rows(v, geom, group, bits, m, k_out, rank > 0)
    P = Q = None
    if rank > 0:
        if P0 is None:
            P0 = draw_p0(B, H, T, D, rank, v.device)
        P, Q = lowrank(err, rank, loop, P0, transposed=False)
    if oidx is not None:
        oidx, oval = oidx.view(B, T, 2 * k_out), oval.view(B, T, 2 * k_out)
    return Payload("v", (B, H, T, D), bits, group, m, code, scale, mn, P, Q, oidx, oval, k_out)


def compress_key_t(kt: torch.Tensor, bits: int, group: int, k_out: int = 0, rank: int = 0, loop: int = 3,
                   mode="fp32", P0: Optional[torch.Tensor] = None) -> Payload:
    """K^T [B,H,D,T] fp16 (what the attention hook passes, modeling_llamagear.py:268) -> Payload
    (per-channel groups along T; outliers per channel row along T)."""
    assert kt.dim() == 4 and kt.dtype == torch.float16
    kt = kt.contiguous()
    L.require_gpu(kt)
    B, H, D, T = kt.shape
    m = _MODES[mode]
    geom = (B * H * D, D, D * T, T, 1, T, 0)
    code, scale, mn, err, oidx, oval = _compress_rows(kt, geom, group, bits, m, k_out, rank > 0)
    P = Q = None
    if rank > 0:
        if P0 is None:
            P0 = draw_p0(B, H, T, D, rank, kt.device)
        P, Q = lowrank(err, rank, loop, P0, transposed=True)     # err is E^T [B,H,D,T]
    if oidx is not None:
        oidx, oval = oidx.view(B, H, D, 2 * k_out), oval.view(B, H, D, 2 * k_out)
    return Payload("k", (B, H, T, D), bits, group, m, code, scale, mn, P, Q, oidx, oval, k_out)


def transpose_last2(x: torch.Tensor) -> torch.Tensor:
    """fp16 [B,H,R,C] -> contiguous [B,H,C,R] with the LDS-tiled HIP kernel (the reference's caller does
    key_states.transpose(2, 3).contiguous(), modeling_llamagear.py:268, :403)."""
    assert x.dim() == 4 and x.dtype == torch.float16
    x = x.contiguous()
    L.require_gpu(x)
    B, H, R, Cc = x.shape
    y = torch.empty((B, H, Cc, R), dtype=torch.float16, device=x.device)
    rc = L.load().gear_transpose_f16(L.ptr(x), B * H, R, Cc, L.ptr(y), L.stream_ptr())
    L.check(rc, "gear_transpose_f16")
    return y


def compress_key(k: torch.Tensor, *args, **kw) -> Payload:
    """K [B,H,T,D] fp16 (token-major) -> Payload (the K^T re-layout runs on the HIP transpose kernel)."""
    return compress_key_t(transpose_last2(k), *args, **kw)


def decompress(p: Payload, transposed_out: bool = False) -> torch.Tensor:
    """Payload -> fp16 [B,H,T,D] (or K^T [B,H,D,T] when transposed_out and kind == 'k')."""
    B, H, T, D = p.shape
    lib = L.load()
    r = p.rank
    k = p.k_out
    if p.kind == "v":
        out = torch.empty((B, H, T, D), dtype=torch.float16, device=p.code.device)
        rc = lib.gear_decompress_rows(L.ptr(p.code), L.ptr(p.scale), L.ptr(p.mn), B * T, T, H * T * D, D, H, D, T * D,
                                      p.group, p.bits, p.mode, 0, L.ptr(p.P), L.ptr(p.Q), r, T, D, L.ptr(p.oidx),
                                      L.ptr(p.oval), k, L.ptr(out), L.stream_ptr())
        L.check(rc, "gear_decompress_rows")
        return out
    out = torch.empty((B, H, D, T), dtype=torch.float16, device=p.code.device)
    rc = lib.gear_decompress_rows(L.ptr(p.code), L.ptr(p.scale), L.ptr(p.mn), B * H * D, D, D * T, T, 1, T, 0, p.group,
                                  p.bits, p.mode, 1, L.ptr(p.P), L.ptr(p.Q), r, T, D, L.ptr(p.oidx), L.ptr(p.oval), k,
                                  L.ptr(out), L.stream_ptr())
    L.check(rc, "gear_decompress_rows")
    return out if transposed_out else transpose_last2(out)
