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

import os

from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib as L

_MODES = {"fp16": 0, "fp32": 1, 0: 0, 1: 1}
# A/B switch for measurements (read once at import): GEAR_KEY_PATH=rows routes compress_key(path="auto") to the round-1 chain
_KEY_PATH_DEFAULT = os.environ.get("GEAR_KEY_PATH", "auto")


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
    oidx: Optional[torch.Tensor] = None      # sorted outlier positions per row and side.  A head SHARD's V payload (compress_value_sharded)
    oval: Optional[torch.Tensor] = None      # pads its lists with index 0xFFFF / value 0: mask with `valid_outliers()` before indexing
    k_out: int = 0

    def valid_outliers(self) -> Optional[torch.Tensor]:
        """bool mask of the list slots that hold an outlier (all of them except the 0xFFFF padding of a head shard's V lists): the
        HIP decompressor and attention kernels skip the padding; a torch-side consumer that scatters oval at oidx must too."""
        return None if self.oidx is None else (self.oidx.to(torch.int32) & 0xFFFF) != 0xFFFF

    @property
    def rank(self):
        return 0 if self.P is None else self.P.shape[-1]

    def chunk_index(self) -> Optional[torch.Tensor]:
        """uint8 chunk index of the sorted outlier lists (built once, cached): K lists (b, h, d, side) x (T/128 + 1) token
        bounds, V lists (b, t, side) x (H + 1) head bounds -- what gear_attn_decode_idx uses to find a chunk's outliers
        without a binary search.  None without outliers or when a list is longer than 255."""
        if self.oidx is None or self.k_out == 0 or self.k_out > 255:
            return None
        if getattr(self, "_ochunk", None) is None:
            B, H, T, D = self.shape
            nb = (T + 127) // 128 + 1 if self.kind == "k" else H + 1
            n_lists = self.oidx.numel() // self.k_out
            out = torch.empty((n_lists, nb), dtype=torch.uint8, device=self.oidx.device)
            rc = L.load().gear_outlier_chunk_index(L.ptr(self.oidx), n_lists, self.k_out, 128, nb, L.ptr(out), L.stream_ptr())
            L.check(rc, "gear_outlier_chunk_index")
            self._ochunk = out
        return self._ochunk

    def nbytes(self) -> int:
        n = 0
        for t in (self.code, self.scale, self.mn, self.P, self.Q, self.oidx, self.oval, getattr(self, "_ochunk", None)):
            if t is not None:
                n += t.numel() * t.element_size()
        return n


def outlier_count(B, H, T, D, sparsity) -> int:
    """k per side per row -- compress_function.py:264-267 / :299-303 (same formula for both layouts, defect B7)."""
    sparsity_num = int(B * H * T * D * sparsity)
    return int(sparsity_num / B / T / 2)


_pin_ring = {}
_p0_queue = []        # bases drawn ahead of their use, in the order they will be asked for: [(key, device, tensor)]


def _draw_p0_now(B, H, S, Dm, rank, device):
    dev = torch.device(device)
    if dev.type != "cuda":
        p = torch.rand(B, H, Dm, rank)
        _ = torch.rand(B, H, S, rank)
        return p.to(dev)
    # through a small ring of PINNED staging buffers with a non-blocking copy: a pageable `.to(device)` stalls the host until the
    # stream has drained
    key = (B, H, Dm, rank)
    ring = _pin_ring.get(key)
    if ring is None:
        ring = _pin_ring[key] = {"bufs": [torch.empty(key, dtype=torch.float32).pin_memory() for _ in range(4)],
                                 "events": [None] * 4, "next": 0}
    i = ring["next"]
    ring["next"] = (i + 1) % 4
    if ring["events"][i] is not None:
        ring["events"][i].synchronize()          # (four draws later: long done)
    buf = ring["bufs"][i]
    torch.rand(key, out=buf)
    _ = torch.rand(B, H, S, rank)
    p = buf.to(dev, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    ring["events"][i] = ev
    return p


def draw_p0(B, H, S, Dm, rank, device):
    """Initial bases exactly as the reference draws them: torch.rand on the CPU generator, P first, then a Q that is
    discarded (new_pack.py:296-297, compress_function.py:83-84) -- keeps the RNG stream aligned with the reference.
    A basis that prefetch_p0() drew ahead (same request, same order) is handed out instead of drawing now."""
    if _p0_queue:
        key, dev, t = _p0_queue[0]
        if key == (B, H, S, Dm, rank) and dev == torch.device(device):
            _p0_queue.pop(0)
            return t
        _p0_queue.clear()       # something else is being compressed (a new prompt): the bases drawn ahead are dropped
    return _draw_p0_now(B, H, S, Dm, rank, device)


def prefetch_p0(specs, device, limit: int = 256):
    """Draw the bases of upcoming draw_p0(B, H, S, Dm, rank) calls NOW, in the given order (the order they will be asked for).
    The values and their order in torch's CPU generator stream are those of drawing at use time -- as long as nothing else draws
    from that generator in between; what moves is the 2 ns per random number of host time (6 ms per decode-time block boundary of a
    32-layer model: two thirds of it the reference's discarded Q draws), from the boundary, where the GPU waits for it, into the
    token steps before it, where the host has slack.  Used by LlamaModel_GEAR's decode loop."""
    for sp in specs:
        if len(_p0_queue) >= limit:
            break
        _p0_queue.append((tuple(sp), torch.device(device), _draw_p0_now(*sp, device)))


def lowrank(E: torch.Tensor, rank: int, loop: int, P0: torch.Tensor, transposed: bool = False, out_dtype=torch.float16,
            out=None):
    """Power iteration on the GPU.  E [B,H,S,Dm] (or [B,H,Dm,S] if transposed), fp16 / fp32.  Returns P [B,H,Dm,r],
    Q [B,H,S,r] (written into `out` = (P, Q) when given)."""
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
    if out is None:
        P = torch.empty((B, H, Dm, rank), dtype=out_dtype, device=E.device)
        Q = torch.empty((B, H, S, rank), dtype=out_dtype, device=E.device)
    else:
        P, Q = out
        assert P.is_contiguous() and Q.is_contiguous() and P.dtype == out_dtype and Q.dtype == out_dtype
    wsb = lib.gear_lowrank_workspace(B * H, S, Dm, rank)
    ws = _workspace(wsb, E.device)
    rc = lib.gear_lowrank(L.ptr(E), 0 if E.dtype == torch.float16 else 1, 1 if transposed else 0, B * H, S, Dm, rank,
                          loop, L.ptr(P0), L.ptr(P), L.ptr(Q), 0 if out_dtype == torch.float16 else 1, L.ptr(ws), wsb,
                          L.stream_ptr())
    L.check(rc, "gear_lowrank")
    return P, Q


def _alloc_rows(shape, n_rows, group, bits, mode, k, dev):
    fpi = 32 // bits
    sdt = torch.float16 if mode == 0 else torch.float32
    code = torch.empty(shape[:-1] + (shape[-1] // fpi,), dtype=torch.int32, device=dev)
    scale = torch.empty(shape[:-1] + (shape[-1] // group,), dtype=sdt, device=dev)
    mn = torch.empty_like(scale)
    oidx = torch.empty((n_rows, 2 * k), dtype=torch.int16, device=dev) if k > 0 else None
    oval = torch.empty((n_rows, 2 * k), dtype=torch.float16, device=dev) if k > 0 else None
    return code, scale, mn, oidx, oval


def _compress_rows(x, geom, group, bits, mode, k, out, err):
    """One gear_compress_rows launch: x (contiguous view) -> out = (code, scale, mn, oidx, oval) views, error into err."""
    n_rows, rows_inner, outer, inner, nseg, seglen, segstride = geom
    code, scale, mn, oidx, oval = out
    rc = L.load().gear_compress_rows(L.ptr(x), n_rows, rows_inner, outer, inner, nseg, seglen, segstride, group, bits,
                                     mode, k, L.ptr(code), L.ptr(scale), L.ptr(mn), L.ptr(err), L.ptr(oidx), L.ptr(oval),
                                     None, L.stream_ptr())
    L.check(rc, "gear_compress_rows")


def compress_rows_once(x, geom, group, bits, mode, k, want_err, err=None):
    """Allocate the outputs and run ONE gear_compress_rows launch over x (benchmarks / profiling tools)."""
    n_rows = geom[0]
    out = _alloc_rows(tuple(x.shape), n_rows, group, bits, mode, k, x.device)
    if want_err and err is None:
        err = torch.empty_like(x)
    _compress_rows(x, geom, group, bits, mode, k, out, err if want_err else None)
    return out + (err,)


def compress_value_sharded(v: torch.Tensor, bits: int, group: int, k_out: int, rank: int, loop: int, mode, P0, shard,
                           thresholds=None) -> Payload:
    """One head shard of a V tensor whose token rows span the heads of `world` ranks: v [B,H_local,T,128], k_out = the FULL row's
    count per side, shard = (tp_rank, tp_world, process group or None).  The selection is the unsharded one (csrc/vsel.hip through
    parallel.exact_v_thresholds: two launches + one all-gather) unless `thresholds` = a ready (thr, fill) pair is passed in (the
    single-process tests and bench.py's emulation of one rank's shard); the rest is compress_value's.  oidx holds LOCAL columns
    (h_local * 128 + d); a row has 2 k_out slots but only the outliers that fall into THIS shard's heads fill them: unused slots
    carry index 0xFFFF and value 0 (see Payload.oidx)."""
    from .parallel import exact_v_thresholds
    tp_rank, tp_world, tp_group = shard
    B, H, T, D = v.shape
    m = _MODES[mode]
    dev = v.device
    code, scale, mn, oidx, oval = _alloc_rows(v.shape, B * T, group, bits, m, k_out, dev)
    P = Q = None
    if rank > 0:
        if P0 is None:
            P0 = draw_p0(B, H, T, D, rank, dev)
        P0 = P0.to(device=dev, dtype=torch.float32).contiguous()
        P = torch.empty((B, H, D, rank), dtype=torch.float16, device=dev)
        Q = torch.empty((B, H, T, rank), dtype=torch.float16, device=dev)
    if thresholds is None and isinstance(tp_group, tuple):       # (rounds 4 - 5 passed the pair in the group's place)
        thresholds, tp_group = tp_group, None
    thr, fill = thresholds if thresholds is not None else exact_v_thresholds(v, k_out, tp_rank, tp_world, tp_group, mode=m)
    lib = L.load()
    wsb = lib.gear_compress_value_fused_workspace(B, H, T, rank)
    ws = _workspace(wsb, dev)
    rc = lib.gear_compress_value_sharded(L.ptr(v), B, H, T, group, bits, m, k_out, L.ptr(code), L.ptr(scale), L.ptr(mn), T, 0, rank,
                                         loop, L.ptr(P0) if rank > 0 else None, L.ptr(P), B * H, 0, L.ptr(Q), T, 0, L.ptr(oidx),
                                         L.ptr(oval), tp_rank * H * D, L.ptr(thr), L.ptr(fill), L.ptr(ws), ws.numel(),
                                         L.stream_ptr(v))
    L.check(rc, "gear_compress_value_sharded")
    return Payload("v", (B, H, T, D), bits, group, m, code, scale, mn, P, Q, oidx.view(B, T, 2 * k_out), oval.view(B, T, 2 * k_out),
                   k_out)


def compress_value(v: torch.Tensor, bits: int, group: int, k_out: int = 0, rank: int = 0, loop: int = 3,
                   mode="fp32", P0: Optional[torch.Tensor] = None, shard=None, thresholds=None) -> Payload:
    """V [B,H,T,D] fp16 -> Payload (per-token groups along D; outliers per token row across heads).
    shard = (tp_rank, tp_world, group): v holds this rank's heads of rows that span `tp_world` ranks; thresholds = a ready
    (thr, fill) pair instead of the exchange -- see compress_value_sharded."""
    assert v.dim() == 4 and v.dtype == torch.float16
    v = v.contiguous()
    L.require_gpu(v)
    if shard is not None and k_out > 0 and shard[1] > 1:
        return compress_value_sharded(v, bits, group, k_out, rank, loop, mode, P0, shard, thresholds)
    B, H, T, D = v.shape
    m = _MODES[mode]
    dev = v.device
    code, scale, mn, oidx, oval = _alloc_rows(v.shape, B * T, group, bits, m, k_out, dev)
    P = Q = err = None
    nb = B
    if rank > 0:
        if P0 is None:
            P0 = draw_p0(B, H, T, D, rank, dev)
        err = torch.empty((nb, H, T, D), dtype=torch.float16, device=dev)
        P = torch.empty((B, H, D, rank), dtype=torch.float16, device=dev)
        Q = torch.empty((B, H, T, rank), dtype=torch.float16, device=dev)
    for b0 in range(0, B, nb):
        b1 = min(B, b0 + nb)
        n = b1 - b0
        geom = (n * T, T, H * T * D, D, H, D, T * D)
        out = (code[b0:b1], scale[b0:b1], mn[b0:b1],
               oidx[b0 * T:b1 * T] if oidx is not None else None, oval[b0 * T:b1 * T] if oval is not None else None)
        _compress_rows(v[b0:b1], geom, group, bits, m, k_out, out, err[:n] if err is not None else None)
        if rank > 0:
            lowrank(err[:n], rank, loop, P0[b0:b1], transposed=False, out=(P[b0:b1], Q[b0:b1]))
    if oidx is not None:
        oidx, oval = oidx.view(B, T, 2 * k_out), oval.view(B, T, 2 * k_out)
    return Payload("v", (B, H, T, D), bits, group, m, code, scale, mn, P, Q, oidx, oval, k_out)


def compress_value_fused(v: torch.Tensor, bits: int, group: int, k_out: int = 0, rank: int = 0, loop: int = 3, mode="fp32",
                         P0: Optional[torch.Tensor] = None) -> Payload:
    """V [B,H,T,128] fp16 -> Payload through ONE gear_compress_value_fused call (csrc/kfused.hip: the chain the streaming cache
    and bench.py run -- row compressor writing the cache geometry, Gram, per-head solve, Q pass) with tcap = T, t_off = 0, so
    the cache tensors ARE the payload tensors.  Same result as compress_value (which drives the leaf entry points one by one)."""
    assert v.dim() == 4 and v.dtype == torch.float16
    v = v.contiguous()
    L.require_gpu(v)
    B, H, T, D = v.shape
    assert D == 128, "gear_compress_value_fused: head_dim 128"
    m = _MODES[mode]
    dev = v.device
    code, scale, mn, oidx, oval = _alloc_rows(v.shape, B * T, group, bits, m, k_out, dev)
    P = Q = None
    if rank > 0:
        if P0 is None:
            P0 = draw_p0(B, H, T, D, rank, dev)
        P0 = P0.to(device=dev, dtype=torch.float32).contiguous()
        P = torch.empty((B, H, D, rank), dtype=torch.float16, device=dev)
        Q = torch.empty((B, H, T, rank), dtype=torch.float16, device=dev)
    lib = L.load()
    ws = _workspace(lib.gear_compress_value_fused_workspace(B, H, T, rank), dev)
    rc = lib.gear_compress_value_fused(L.ptr(v), B, H, T, group, bits, m, k_out, L.ptr(code), L.ptr(scale), L.ptr(mn), T, 0, rank,
                                       loop, L.ptr(P0) if rank > 0 else None, L.ptr(P), B * H, 0, L.ptr(Q), T, 0, L.ptr(oidx),
                                       L.ptr(oval), L.ptr(ws), ws.numel(), L.stream_ptr(v))
    L.check(rc, "gear_compress_value_fused")
    if oidx is not None:
        oidx, oval = oidx.view(B, T, 2 * k_out), oval.view(B, T, 2 * k_out)
    return Payload("v", (B, H, T, D), bits, group, m, code, scale, mn, P, Q, oidx, oval, k_out)


def key_fused_supported(T: int, D: int, group: int, bits: int, k_out: int) -> bool:
    """Shapes the fused token-major K path (gear_compress_key_fused) covers; everything else takes the row compressor."""
    return D == 128 and T % 64 == 0 and 64 <= T <= 16384 and group in (32, 64) and bits in (2, 4) and 2 * k_out <= T


_ws_cache = {}


def _workspace(nbytes: int, dev) -> torch.Tensor:
    """One grow-only scratch buffer per device and stream for the fused compress calls (the work in it is stream-ordered)."""
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = None
        _ws_cache[key] = None
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        _ws_cache[key] = ws
    return ws


def compress_key_fused(k: torch.Tensor, bits: int, group: int, k_out: int = 0, rank: int = 0, loop: int = 3, mode="fp32",
                       P0: Optional[torch.Tensor] = None, variant: int = 0) -> Payload:
    """K [B,H,T,128] fp16 token-major -> Payload through the fused path (csrc/kfused.hip): select -> fused quantize + pack +
    error + Gram -> per-head solve -> Q pass.  No K^T, no transpose kernel."""
    assert k.dim() == 4 and k.dtype == torch.float16
    k = k.contiguous()
    L.require_gpu(k)
    B, H, T, D = k.shape
    if not key_fused_supported(T, D, group, bits, k_out):
        raise L.GearError(f"compress_key_fused: unsupported shape T={T} D={D} group={group} bits={bits} k={k_out}")
    m = _MODES[mode]
    dev = k.device
    fpi = 32 // bits
    sdt = torch.float16 if m == 0 else torch.float32
    code = torch.empty((B, H, D, T // fpi), dtype=torch.int32, device=dev)
    scale = torch.empty((B, H, D, T // group), dtype=sdt, device=dev)
    mn = torch.empty_like(scale)
    oidx = torch.empty((B, H, D, 2 * k_out), dtype=torch.int16, device=dev) if k_out > 0 else None
    oval = torch.empty((B, H, D, 2 * k_out), dtype=torch.float16, device=dev) if k_out > 0 else None
    P = Q = None
    if rank > 0:
        if P0 is None:
            P0 = draw_p0(B, H, T, D, rank, dev)
        P0 = P0.to(device=dev, dtype=torch.float32).contiguous()
        assert tuple(P0.shape) == (B, H, D, rank)
        P = torch.empty((B, H, D, rank), dtype=torch.float16, device=dev)
        Q = torch.empty((B, H, T, rank), dtype=torch.float16, device=dev)
    lib = L.load()
    wsb = lib.gear_compress_key_fused_workspace(B * H, T, k_out, rank)
    ws = _workspace(wsb, dev)
    rc = lib.gear_compress_key_fused(L.ptr(k), B * H, T, group, bits, m, k_out, L.ptr(code), L.ptr(scale), L.ptr(mn),
                                     T // fpi, T // group, 0, rank, loop, L.ptr(P0) if rank > 0 else None, L.ptr(P), B * H, 0,
                                     L.ptr(Q), T, 0, L.ptr(oidx), L.ptr(oval), k_out, 0, variant, L.ptr(ws), ws.numel(),
                                     L.stream_ptr(k))
    L.check(rc, "gear_compress_key_fused")
    return Payload("k", (B, H, T, D), bits, group, m, code, scale, mn, P, Q, oidx, oval, k_out)


def _compress_key_impl(src: torch.Tensor, src_is_transposed: bool, bits, group, k_out, rank, loop, mode, P0) -> Payload:
    assert src.dim() == 4 and src.dtype == torch.float16
    src = src.contiguous()
    L.require_gpu(src)
    if src_is_transposed:
        B, H, D, T = src.shape
    else:
        B, H, T, D = src.shape
    m = _MODES[mode]
    dev = src.device
    code, scale, mn, oidx, oval = _alloc_rows((B, H, D, T), B * H * D, group, bits, m, k_out, dev)
    P = Q = err = ktb = None
    nb = B
    if rank > 0:
        if P0 is None:
            P0 = draw_p0(B, H, T, D, rank, dev)
        err = torch.empty((nb, H, D, T), dtype=torch.float16, device=dev)   # E^T
        P = torch.empty((B, H, D, rank), dtype=torch.float16, device=dev)
        Q = torch.empty((B, H, T, rank), dtype=torch.float16, device=dev)
    if not src_is_transposed:
        ktb = torch.empty((nb, H, D, T), dtype=torch.float16, device=dev)   # K^T
    lib = L.load()
    for b0 in range(0, B, nb):
        b1 = min(B, b0 + nb)
        n = b1 - b0
        if src_is_transposed:
            kt = src[b0:b1]
        else:
            kt = ktb[:n]
            L.check(lib.gear_transpose_f16(L.ptr(src[b0:b1]), n * H, T, D, L.ptr(kt), L.stream_ptr()), "gear_transpose_f16")
        geom = (n * H * D, D, D * T, T, 1, T, 0)
        r0, r1 = b0 * H * D, b1 * H * D
        out = (code[b0:b1], scale[b0:b1], mn[b0:b1],
               oidx[r0:r1] if oidx is not None else None, oval[r0:r1] if oval is not None else None)
        _compress_rows(kt, geom, group, bits, m, k_out, out, err[:n] if err is not None else None)
        if rank > 0:
            lowrank(err[:n], rank, loop, P0[b0:b1], transposed=True, out=(P[b0:b1], Q[b0:b1]))   # err is E^T [n,H,D,T]
    if oidx is not None:
        oidx, oval = oidx.view(B, H, D, 2 * k_out), oval.view(B, H, D, 2 * k_out)
    return Payload("k", (B, H, T, D), bits, group, m, code, scale, mn, P, Q, oidx, oval, k_out)


def compress_key_t(kt: torch.Tensor, bits: int, group: int, k_out: int = 0, rank: int = 0, loop: int = 3,
                   mode="fp32", P0: Optional[torch.Tensor] = None) -> Payload:
    """K^T [B,H,D,T] fp16 (what the attention hook passes, modeling_llamagear.py:268) -> Payload
    (per-channel groups along T; outliers per channel row along T)."""
    return _compress_key_impl(kt, True, bits, group, k_out, rank, loop, mode, P0)


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


def compress_key(k: torch.Tensor, bits: int, group: int, k_out: int = 0, rank: int = 0, loop: int = 3,
                 mode="fp32", P0: Optional[torch.Tensor] = None, path: str = "auto") -> Payload:
    """K [B,H,T,D] fp16 (token-major) -> Payload.  path "auto": the fused token-major path (compress_key_fused) when the
    shape allows it (head_dim 128, T % 64 == 0, group 32 / 64), otherwise -- or with path "rows" -- the round-1 chain
    K^T re-layout -> row compressor -> Gram -> Q pass; "fused" insists on the fused path."""
    assert path in ("auto", "rows", "fused")
    if path == "auto" and _KEY_PATH_DEFAULT != "auto":
        path = _KEY_PATH_DEFAULT
    B, H, T, D = k.shape
    if path == "fused" or (path == "auto" and key_fused_supported(T, D, group, bits, k_out)):
        return compress_key_fused(k, bits, group, k_out, rank, loop, mode, P0)
    return _compress_key_impl(k, False, bits, group, k_out, rank, loop, mode, P0)


def quant_whole_rows(x: torch.Tensor, layout: str, bits: int, mode="fp32", oidx: Optional[torch.Tensor] = None, k_out: int = 0,
                     want_err: bool = False):
    """Quantize -> dequantize with ONE group per row (the KCVT variants, csrc/rows_whole.hip).  x fp16 [B,H,T,D].
    layout "k": a row = one channel over all T tokens (group = seq_len); "v": a row = one token across all heads (group =
    H*D).  oidx: the rows' outlier lists as compress_key / compress_value return them (they keep their original value).
    Returns y fp16 [B,H,T,D] (and the error x - y when want_err)."""
    assert x.dim() == 4 and x.dtype == torch.float16 and layout in ("k", "v")
    x = x.contiguous()
    L.require_gpu(x, oidx)
    B, H, T, D = x.shape
    y = torch.empty_like(x)
    err = torch.empty_like(x) if want_err else None
    if layout == "k":     # row (bh, d): elements x[bh, t, d], t = 0..T-1 -> T segments of one element, D apart
        geom = (B * H * D, D, T * D, 1, T, 1, D)
    else:                 # row (b, t): H segments of D contiguous elements, T*D apart
        geom = (B * T, T, H * T * D, D, H, D, T * D)
    rc = L.load().gear_quant_rows_whole(L.ptr(x), *geom, bits, _MODES[mode], L.ptr(oidx), k_out if oidx is not None else 0,
                                        L.ptr(y), L.ptr(err), L.stream_ptr(x))
    L.check(rc, "gear_quant_rows_whole")
    return (y, err) if want_err else y


def quant_rows_ragged(x: torch.Tensor, layout: str, bits: int, group: int, k_out: int = 0, mode="fp32", want_err: bool = False):
    """gears_channelQ / gears_tokenQ on rows of any length (csrc/rows_ragged.hip): selection of the k_out smallest / largest and
    the fill mean over the WHOLE row, quantization of the first floor(len / group) * group elements in groups of `group`, the
    tail untouched (compress_function.py:107-122, :261-333).  x fp16 [B,H,T,D]; layout "k": a row = one channel over the T
    tokens, "v": a row = one token across the heads.  Returns y fp16 [B,H,T,D] (and the error x - y when want_err)."""
    assert x.dim() == 4 and x.dtype == torch.float16 and layout in ("k", "v")
    x = x.contiguous()
    L.require_gpu(x)
    B, H, T, D = x.shape
    y = torch.empty_like(x)
    err = torch.empty_like(x) if want_err else None
    if layout == "k":
        geom = (B * H * D, D, T * D, 1, T, 1, D)
    else:
        geom = (B * T, T, H * T * D, D, H, D, T * D)
    rc = L.load().gear_quant_rows_ragged(L.ptr(x), *geom, group, bits, _MODES[mode], k_out, L.ptr(y), L.ptr(err), L.stream_ptr(x))
    L.check(rc, "gear_quant_rows_ragged")
    return (y, err) if want_err else y


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
