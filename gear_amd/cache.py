"""Pre-allocated, kernel-native GEAR KV cache for one attention layer (the build's own
design: the reference rebuilds every payload tensor with torch.cat on each block boundary,
cuda_supported_gear/modeling_llamagear.py:273-286, :365-378, and re-lays the whole packed K out on every token,
quant/matmul.py:205, :215-216).

State machine = the attention hook's (modeling_llamagear.py:177-484): an fp16 window of the most recent < `residual`
tokens; when it fills, the block is compressed IN PLACE behind the already compressed tokens: quantized backbone, per-block
rank-r factors and -- when the config carries a sparsity (`left`, the simulated path's name,
GenerationBench/.../Simulated/compress_config.py) -- the sparse outliers of the block, which is what the reference's
streaming hook applies to every new block (Simulated/modeling_llama_new.py:979-1019 -> compress_function.py:261-333).
Layout = what gear_attn_decode_cache streams: K channel-major with a fixed row pitch (a block append is 128 short row
segments, never a re-layout), V token-major, token-side factors per token, channel-side factors per segment (segment 0 =
the prompt, then one per block), K outlier lists per (channel, side) that grow by `kk_blk` entries per block (later blocks
hold later tokens, so the lists stay sorted), V outlier lists per token row.

The compress calls write straight into these tensors (gear_compress_key_fused / gear_compress_value_fused with a token
offset and row pitches): no intermediate payload, no strided copies.

Outlier counts.  V rows (a token across the heads) and the K rows of the prompt segment use the reference's formula
(compress_function.py:264-267, :299-303: int(H*D*s/2) per side, capped at half the row).  For the K rows of a 64-token
decode block that formula asks for more outliers than the row has elements (defect B7: it depends on H*D, not on the
row length; torch.topk then takes overlapping sets and the block ends up stored losslessly); the cache uses the NOMINAL
count max(1, round(64*s/2)) per side there (`block_outlier_count="reference"` in the config selects the reference's count, capped at
half the row).

Differences from the hook, on purpose: K and V are compressed in lockstep (the hook compresses V only when T > residual,
:416, which strands V in fp16 when the prompt is exactly `residual` long), and the block factors start from a
channel-side random basis (the simulated path's orientation, compress_function.py:83) so that the Gram-matrix kernel
applies; both are rank-r power-iteration approximations of the same error matrix.
"""
from __future__ import annotations

import ctypes as C_
import math

import torch

from . import _lib as L
from . import compress as C


def _sparsity(cc) -> float:
    return float(cc.get("left", cc.get("sparsity", 0.0)) or 0.0)


def _cache_dims(batch, n_kv_heads, max_tokens, cc, head_dim=128, heads_total=None, v_exact=False):
    bits, group, R = cc["quantize_bit"], cc["group_size"], cc["residual"]
    m = cc["compress_method"]
    lowrank = ("gearl" in m) or ("gearsl" in m)
    rk = int(cc["rank"]) if lowrank else 0
    rv = int(cc["rankv"]) if lowrank else 0
    if R not in (64, 128):
        raise L.GearError(f"GearKVCache: residual must be 64 or 128 (got {R}): the decode attention kernel holds at most 128 "
                          "window tokens (gear_attn_decode: 0 <= W <= 128) and blocks are whole 64-token tiles")
    assert R % group == 0 and group in (32, 64) and bits in (2, 4), "GearKVCache: group 32 / 64, 2 or 4 bits"
    fpi = 32 // bits
    Tmax = (max_tokens + R - 1) // R * R
    assert Tmax <= 16384, "GearKVCache: at most 16384 tokens"
    nseg = 1 + Tmax // R
    B, H, D, T = batch, n_kv_heads, head_dim, Tmax
    s = _sparsity(cc)
    kv = int(int(B * H * T * D * s) / B / T / 2) if s > 0 else 0          # per side per token row (compress_function.py:300-303)
    # the same formula for the prompt's channel rows -- on the FULL head count when this cache is one head shard of several
    # (a K row lives inside one head, so its outlier count must not depend on how the heads are spread over GPUs; a V row
    # spans the heads, so a shard selects k / world inside its own heads: see parallel.py)
    Ht = heads_total or H
    if v_exact and Ht != H:
        # exact cross-shard V selection (parallel.exact_v_selection): the row's outliers are chosen over ALL heads, a shard keeps
        # the ones that fall into its heads -- anything between 0 and the full count, so its lists hold the full count per side
        kv = int(int(B * Ht * T * D * s) / B / T / 2) if s > 0 else 0
        if kv > H * D:
            raise L.GearError(f"GearKVCache: {kv} V outliers per side and row exceed the shard's row length {H * D} (sparsity > 1 / "
                              "world): not a configuration the exact cross-shard selection supports; use v_selection='per_shard'")
    kk0_max = int(int(B * Ht * T * D * s) / B / T / 2) if s > 0 else 0
    # K outliers per side and channel row of a decode block: "nominal" = the sparsity applied to the row's own length; "reference" =
    # the reference's formula (B7: it depends on H*D, not on the row length), capped at half the row -- what gears_channelQ does to a
    # 64-token block with the streaming hook's config (compress_function.py:264-267): at s = 0.02 on 7B that stores the block losslessly
    if cc.get("block_outlier_count", "nominal") == "reference":
        kk_blk = min(int(int(B * Ht * R * D * s) / B / R / 2), R // 2) if s > 0 else 0
    else:
        kk_blk = max(1, round(R * s / 2)) if s > 0 else 0                 # nominal count for a 64-token block (B7)
    kcap = kk0_max + (Tmax // R) * kk_blk
    shapes = dict(kcode=((B, H, D, T // fpi), torch.int32), kscale=((B, H, D, T // group), torch.float16),
                  kmn=((B, H, D, T // group), torch.float16), vcode=((B, H, T, D // fpi), torch.int32),
                  vscale=((B, H, T, D // group), torch.float16), vmn=((B, H, T, D // group), torch.float16),
                  kwin=((B, H, R, D), torch.float16), vwin=((B, H, R, D), torch.float16))
    if lowrank:
        shapes.update(kPseg=((nseg, B, H, D, rk), torch.float16), kQtok=((B, H, T, rk), torch.float16),
                      vPseg=((nseg, B, H, D, rv), torch.float16), vQtok=((B, H, T, rv), torch.float16))
    nbk = Tmax // 128 + 2                 # pitch of the K chunk index: bounds 0, 128, ... over the prompt segment
    if kk_blk > 0:
        shapes.update(koidx=((B, H, D, 2, kcap), torch.int16), koval=((B, H, D, 2, kcap), torch.float16))
        if kk0_max <= 255:
            shapes.update(kochunk=((B, H, D, 2, nbk), torch.uint8))
    if kv > 0:
        shapes.update(voidx=((B, T, 2 * kv), torch.int16), voval=((B, T, 2 * kv), torch.float16))
        if kv <= 255:
            shapes.update(vochunk=((B, T, 2, H + 1), torch.uint8))
    # sparse tiles (gear_cache_tiles_build): the outlier corrections of a 128-token K chunk / 64-token V block as a flat list,
    # sized for the expected count plus slack; a chunk that overflows falls back to the lists (count -1)
    nck, nblk = Tmax // 128 + 1, 2 * (Tmax // 128 + 1)
    ktile_cap = vtile_cap = 0
    if kk_blk > 0:
        per_chunk = D * 2 * max(2 * kk_blk, math.ceil(128 * kk0_max / max(Tmax, 1)))
        ktile_cap = (int(per_chunk * 1.25) + 64 + 255) // 256 * 256
        ktile_cap = max(ktile_cap, 512)
        shapes.update(ktile=((B, H, nck, ktile_cap), torch.int32), kcnt=((B, H, nck), torch.int32))
    if kv > 0:
        per_blk = 64 * 2 * kv / H
        vtile_cap = max(256, (int(per_blk * 1.25) + 32 + 255) // 256 * 256)
        shapes.update(vtile=((B, H, nblk, vtile_cap), torch.int32), vcnt=((B, H, nblk), torch.int32))
    return shapes, dict(bits=bits, group=group, R=R, lowrank=lowrank, rk=rk, rv=rv, fpi=fpi, Tmax=Tmax, nseg=nseg,
                        kv=kv, kk0_max=kk0_max, kk_blk=kk_blk, kcap=kcap, nbk=nbk, nck=nck, nblk=nblk, ktile_cap=ktile_cap,
                        vtile_cap=vtile_cap)


def _view(bufs, d, NB, H, D, kk0, seg0, kwin=True):
    """The gear_cache_view of a set of cache tensors (NB = leading batch dimension: layers * batch for pooled storage)."""
    v = L.CacheView()
    p = L.ptr
    for f, n in (("kcode", "kcode"), ("kscale", "kscale"), ("kmn", "kmn"), ("kP", "kPseg"), ("kQ", "kQtok"), ("koidx", "koidx"),
                 ("koval", "koval"), ("vcode", "vcode"), ("vscale", "vscale"), ("vmn", "vmn"), ("vP", "vPseg"), ("vQ", "vQtok"),
                 ("voidx", "voidx"), ("voval", "voval"), ("vochunk", "vochunk"), ("ktile", "ktile"), ("kcnt", "kcnt"),
                 ("vtile", "vtile"), ("vcnt", "vcnt")):
        setattr(v, f, p(bufs.get(n)))
    v.kochunk = p(bufs.get("kochunk")) if kk0 else None
    if kwin:
        v.kwin, v.vwin = p(bufs["kwin"]), p(bufs["vwin"])
    v.B, v.Hkv, v.D, v.tcap = NB, H, D, d["Tmax"]
    v.ldk, v.lsk, v.group, v.bits, v.mode = d["Tmax"] // d["fpi"], d["Tmax"] // d["group"], d["group"], d["bits"], 0
    v.rk, v.rv = d["rk"], d["rv"]
    v.kk_cap, v.kk0, v.kkb, v.kv = (d["kcap"] if d["kk_blk"] else 0), kk0, d["kk_blk"], d["kv"]
    v.seg0, v.seglen, v.wcap = seg0, (d["R"] if (d["lowrank"] or d["kk_blk"]) else 0), d["R"]
    v.nbk_pitch, v.ktile_cap, v.nck, v.vtile_cap, v.nblk = d["nbk"], d["ktile_cap"], d["nck"], d["vtile_cap"], d["nblk"]
    return v


USE_BLOCK_KERNEL = True      # tests switch it off to compare the single-launch block compressor with the chain
# With V outliers every K tile's wave first selects 64 / (2 H) token rows (one after the other): at 2 - 3 KV heads per rank that
# serial part outweighs the saved launches (measured in round 3 at 80 layers x 1 head: 367 us against the chain's 222).  ONE KV head
# per rank (70B on 8 GPUs) needs no hand-off at all -- the row is the tile's own -- and takes the block kernel (round 4).
BLOCK_KERNEL_MIN_HEADS = 4
_BLOCK_WS = {}


def _block_sync_ws(lib, NB, H, dev):
    """Hand-off buffer of gear_compress_block (row flags, masks, means): zeroed once, then owned by the kernel."""
    key = (str(dev), NB, H)
    ws = _BLOCK_WS.get(key)
    if ws is None:
        ws = _BLOCK_WS[key] = torch.zeros((lib.gear_compress_block_workspace(NB, H),), dtype=torch.uint8, device=dev)
    return ws


def block_kernel_status(dev="cuda") -> int:
    """OR of the status words of every block-compress hand-off buffer on `dev` (non-zero: a V tile gave up waiting)."""
    st = 0
    want = torch.device(dev)
    if want.type == "cuda" and want.index is None:
        want = torch.device("cuda", torch.cuda.current_device())
    for ws in _BLOCK_WS.values():
        if ws.device == want:
            off = (-ws.data_ptr()) % 256
            st |= int(ws[off:off + 4].view(torch.int32).item())
    return st


def _draw_p0(shape_local, gen, dev, tp):
    """Random start bases [NB, H_local, D, r].  Head-sharded: every rank draws the bases of ALL heads from the same stream and keeps
    its own (so that a head's factors do not depend on how the heads are spread over GPUs)."""
    NB, H, D, r = shape_local
    if tp is None or tp["world"] == 1:
        return torch.rand(shape_local, device=dev, generator=gen)
    full = torch.rand((NB, H * tp["world"], D, r), device=dev, generator=gen)
    return full[:, tp["rank"] * H:(tp["rank"] + 1) * H].contiguous()


def _compress_value_exact(bufs, d, lead, B, H, D, v_src, T, t_off, seg, loop, P0v, tp, kk0, seg0):
    """V payload of a head shard with the outliers selected over the WHOLE token row (all ranks' heads), bit-identical to the matching
    head slice of the unsharded payload (tests/test_gpu_parallel.py).  Round 5: in HIP -- gear_vsel_candidates, ONE all-gather of
    4 (2 kv + 2) bytes per row and rank, gear_vsel_thresholds, then gear_compress_value_sharded (the row compressor with the selection
    given + the usual low-rank step) writing straight behind token t_off of the cache: 3 launches + the chain's low-rank kernels where
    round 4 ran ~25 torch launches (parallel.exact_v_selection: two topk, where, gather ...; kept as the tests' cross-check with
    tp["exact"] == "torch")."""
    from .parallel import exact_v_selection, exact_v_thresholds
    lib = L.load()
    p = L.ptr
    NB = lead * B
    Tmax, g, bits, kv = d["Tmax"], d["group"], d["bits"], d["kv"]
    st = L.stream_ptr(v_src)
    if tp.get("exact", True) == "torch":
        from .quant.new_pack import triton_quantize_and_pack_along_last_dim_witherror
        v4 = v_src.reshape(NB, H, -1, D)[:, :, :T]
        filled, mask, oidx, oval = exact_v_selection(v4, kv, tp["rank"], tp["world"], tp.get("group"))
        code, scale, mn, err = triton_quantize_and_pack_along_last_dim_witherror(filled, g, bits)
        bufs["vcode"].view(NB, H, Tmax, -1)[:, :, t_off:t_off + T] = code
        bufs["vscale"].view(NB, H, Tmax, -1)[:, :, t_off:t_off + T] = scale
        bufs["vmn"].view(NB, H, Tmax, -1)[:, :, t_off:t_off + T] = mn
        bufs["voidx"].view(NB, Tmax, 2 * kv)[:, t_off:t_off + T] = oidx
        bufs["voval"].view(NB, Tmax, 2 * kv)[:, t_off:t_off + T] = oval
        if d["lowrank"]:
            err = err.view(NB, H, T, D).masked_fill_(mask, 0)
            P, Q = C.lowrank(err, d["rv"], loop, P0v)
            bufs["vPseg"].view(lead, d["nseg"], B, H, D, d["rv"])[:, seg] = P.view(lead, B, H, D, d["rv"])
            bufs["vQtok"].view(NB, H, Tmax, d["rv"])[:, :, t_off:t_off + T] = Q
    else:
        v4 = v_src.reshape(NB, H, -1, D)
        if v4.shape[2] != T:
            v4 = v4[:, :, :T].contiguous()
        thr, fill = exact_v_thresholds(v4, kv, tp["rank"], tp["world"], tp.get("group"), mode=0)
        seg_v = B * H * D * d["rv"]
        vP = bufs["vPseg"].view(-1)[seg * seg_v:] if d["lowrank"] else None
        wsb = lib.gear_compress_value_fused_workspace(NB, H, T, d["rv"])
        ws = C._workspace(wsb, v_src.device)
        rc = lib.gear_compress_value_sharded(
            p(v4), NB, H, T, g, bits, 0, kv, p(bufs["vcode"]), p(bufs["vscale"]), p(bufs["vmn"]), Tmax, t_off, d["rv"], loop, p(P0v),
            p(vP), B * H, d["nseg"] * seg_v, p(bufs.get("vQtok")), Tmax, t_off, p(bufs["voidx"]), p(bufs["voval"]),
            tp["rank"] * H * D, p(thr), p(fill), p(ws), ws.numel(), st)
        L.check(rc, "gear_compress_value_sharded")
    if "vochunk" in bufs:     # (the sentinel index 0xFFFF of an unused list slot lies beyond every head bound: the terminal entry is the count)
        rc = lib.gear_outlier_chunk_index_ex(p(bufs["voidx"]), NB, 2 * T, 2 * Tmax, 2 * t_off, kv, kv, 128, H + 1,
                                             p(bufs["vochunk"]), H + 1, st)
        L.check(rc, "gear_outlier_chunk_index_ex(V)")


def _compress_into(bufs, d, lead, B, H, D, k_src, v_src, T, t_off, seg, kk, o_off, loop, gen, kk0=0, seg0=0, tp=None):
    """Compress K / V [lead*B, H, T, 128] (lead = layers riding in the batch dimension of pooled storage) and write the
    payload behind token t_off of the cache tensors in `bufs` (same leading dimension), factors into segment `seg`, K outlier
    lists at position o_off.  fp16-stepwise arithmetic (the fused path's mode)."""
    lib = L.load()
    p = L.ptr
    NB = lead * B
    dev = k_src.device
    Tmax, g, fpi, bits = d["Tmax"], d["group"], d["fpi"], d["bits"]
    P0k = P0v = None
    if d["lowrank"]:
        P0k = _draw_p0((NB, H, D, d["rk"]), gen, dev, tp)
        P0v = _draw_p0((NB, H, D, d["rv"]), gen, dev, tp)
    v_exact = tp is not None and tp["world"] > 1 and tp.get("exact", True) and d["kv"] > 0
    if (USE_BLOCK_KERNEL and not v_exact and T == d["R"] == 64 and k_src is bufs.get("kwin") and v_src is bufs.get("vwin") and H <= 64
            and kk <= 16 and (d["kv"] <= 255 or "vochunk" not in bufs) and (H >= BLOCK_KERNEL_MIN_HEADS or H == 1 or not d["kv"])):
        # the decode-time block boundary: ONE launch over all (layer, head, K | V) tiles (csrc/block_fused.hip) instead of the
        # chain below (select, fused quantize + Gram, solve, Q pass; row compressor, Gram + solve, Q pass; chunk index; 2 tile
        # builders).  Same payload bits; the factors come from the token-side iteration (same subspace).
        view = _view(bufs, d, NB, H, D, kk0, seg0, kwin=True)
        if not kk:
            view.koidx = view.koval = None
        seg_k, seg_v = B * H * D * d["rk"], B * H * D * d["rv"]
        kP = bufs["kPseg"].view(-1)[seg * seg_k:] if d["lowrank"] else None
        vP = bufs["vPseg"].view(-1)[seg * seg_v:] if d["lowrank"] else None
        ws = _block_sync_ws(lib, NB, H, dev)
        rc = lib.gear_compress_block(C_.byref(view), t_off, o_off, loop, p(P0k), p(P0v), p(kP), p(vP), B * H, d["nseg"] * seg_k,
                                     d["nseg"] * seg_v, p(ws), ws.numel(), L.stream_ptr(k_src))
        L.check(rc, "gear_compress_block")
        return
    # per-segment channel factors [lead, nseg, B, H, D, r]: head (l, b, h) of segment seg
    def pseg(name, r):
        if not d["lowrank"]:
            return None, 0
        t = bufs[name]
        seg_elems = B * H * D * r
        return t.view(-1)[seg * seg_elems:], d["nseg"] * seg_elems
    kP, kP_stride = pseg("kPseg", d["rk"])
    vP, vP_stride = pseg("vPseg", d["rv"])
    wsb = max(lib.gear_compress_key_fused_workspace(NB * H, T, kk, d["rk"]),
              lib.gear_compress_value_fused_workspace(NB, H, T, d["rv"]))
    ws = C._workspace(wsb, dev)
    st = L.stream_ptr(k_src)
    rc = lib.gear_compress_key_fused(
        p(k_src), NB * H, T, g, bits, 0, kk, p(bufs["kcode"]), p(bufs["kscale"]), p(bufs["kmn"]), Tmax // fpi, Tmax // g,
        t_off, d["rk"], loop, p(P0k), p(kP), B * H, kP_stride, p(bufs.get("kQtok")), Tmax, t_off,
        p(bufs.get("koidx")) if kk else None, p(bufs.get("koval")) if kk else None, d["kcap"], o_off, 0, p(ws), ws.numel(), st)
    L.check(rc, "gear_compress_key_fused")
    if v_exact:
        _compress_value_exact(bufs, d, lead, B, H, D, v_src, T, t_off, seg, loop, P0v, tp, kk0, seg0)
    else:
        rc = lib.gear_compress_value_fused(
            p(v_src), NB, H, T, g, bits, 0, d["kv"], p(bufs["vcode"]), p(bufs["vscale"]), p(bufs["vmn"]), Tmax, t_off, d["rv"],
            loop, p(P0v), p(vP), B * H, vP_stride, p(bufs.get("vQtok")), Tmax, t_off, p(bufs.get("voidx")), p(bufs.get("voval")),
            p(ws), ws.numel(), st)
        L.check(rc, "gear_compress_value_fused")
    # chunk indices of the sparse lists (what lets a 128-token attention chunk find its outliers without a search):
    # V: one row of H + 1 head bounds per new (token row, side); K: the prompt segment's entries, once, at prefill
    if d["kv"] and "vochunk" in bufs and not v_exact:
        rc = lib.gear_outlier_chunk_index_ex(p(bufs["voidx"]), NB, 2 * T, 2 * Tmax, 2 * t_off, d["kv"], d["kv"], 128, H + 1,
                                             p(bufs["vochunk"]), H + 1, st)
        L.check(rc, "gear_outlier_chunk_index_ex(V)")
    if kk and seg == 0 and t_off == 0 and "kochunk" in bufs:
        rc = lib.gear_outlier_chunk_index_ex(p(bufs["koidx"]), 1, NB * H * D * 2, 0, 0, kk, d["kcap"], 128, (T + 127) // 128 + 1,
                                             p(bufs["kochunk"]), d["nbk"], st)
        L.check(rc, "gear_outlier_chunk_index_ex(K)")
    if "ktile" in bufs or "vtile" in bufs:      # sparse tiles of the chunks / blocks the new tokens lie in
        view = _view(bufs, d, NB, H, D, kk0, seg0, kwin=False)
        rc = lib.gear_cache_tiles_build(C_.byref(view), t_off + T, t_off // 128, (t_off + T + 127) // 128, t_off // 64,
                                        (t_off + T) // 64, st)
        L.check(rc, "gear_cache_tiles_build")


class GearKVCachePool:
    """The buffers of ALL layers' caches as one tensor per field with a leading layer dimension.  Every layer's
    GearKVCache takes its (contiguous) slice, so nothing changes for the kernels; what the pool buys is the block boundary:
    all layers fill their fp16 windows on the same token, and with pooled storage the 32 per-layer compress sequences become
    ONE gear_compress_key_fused + ONE gear_compress_value_fused over [layers * batch, H, residual, 128] that write the block
    of every layer in place (compress_all): 7 launches per block boundary."""

    def __init__(self, n_layers: int, batch: int, n_kv_heads: int, max_tokens: int, compress_config: dict, device,
                 head_dim: int = 128, seed: int = 0, heads_total: int = None, tp: dict = None):
        """tp = dict(rank, world, group[, exact=True]): this pool holds one head shard of `world`; with exact (default) the V outliers
        of a token row are selected over all ranks' heads (one small all-gather per compress call, parallel.exact_v_selection)
        instead of k / world inside the shard."""
        self.tp = tp
        v_exact = tp is not None and tp["world"] > 1 and tp.get("exact", True)
        shapes, self.dims = _cache_dims(batch, n_kv_heads, max_tokens, compress_config, head_dim, heads_total, v_exact)
        if batch * n_kv_heads > 65535:
            raise L.GearError(f"GearKVCachePool: batch * kv heads = {batch * n_kv_heads} exceeds 65535 (one layer's heads ride on a "
                              "grid dimension of the prefill kernels)")
        self.L, self.B, self.H, self.D = n_layers, batch, n_kv_heads, head_dim
        self.loop = int(compress_config.get("loop", 3))
        self.buf = {}
        for n, (shp, dt) in shapes.items():
            if n in ("kPseg", "vPseg"):
                self.buf[n] = torch.zeros((n_layers,) + shp, dtype=dt, device=device)     # [L, nseg, B, H, D, r]
            else:
                self.buf[n] = torch.zeros((n_layers,) + shp, dtype=dt, device=device)
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(seed)
        self.caches = []

    def compress_all(self):
        """Every layer's window holds `residual` tokens: compress all of them in one go, in place behind the compressed
        tokens (the layers ride in the batch dimension)."""
        d, b, c0 = self.dims, self.buf, self.caches[0]
        R = d["R"]
        assert all(c.n_win == R and c.n_comp == c0.n_comp and c.kk0 == c0.kk0 for c in self.caches)
        t0, seg = c0.n_comp, c0._segment_of(c0.n_comp)
        assert t0 + R <= d["Tmax"], "cache capacity exceeded"
        o_off = c0.kk0 + ((t0 - c0.seg0) // R) * d["kk_blk"]
        # the chain's kernels put (layer, batch, head) on grid.y (<= 65535): when the single-launch block kernel does not apply
        # (see _compress_into) and the pool is larger than that, the layers go in groups
        per = max(1, 65535 // (self.B * self.H))
        groups = [(0, self.L)] if self.L <= per else [(l0, min(self.L, l0 + per)) for l0 in range(0, self.L, per)]
        for l0, l1 in groups:
            sub = b if (l0, l1) == (0, self.L) else {n: t[l0:l1] for n, t in b.items()}
            _compress_into(sub, d, l1 - l0, self.B, self.H, self.D, sub["kwin"], sub["vwin"], R, t0, seg, d["kk_blk"], o_off,
                           self.loop, self.gen, c0.kk0, c0.seg0, self.tp)
        for c in self.caches:
            c.n_comp += R
            c.n_win = 0


class GearKVCache:
    def __init__(self, batch: int, n_kv_heads: int, max_tokens: int, compress_config: dict, device, head_dim: int = 128,
                 seed: int = 0, state: torch.Tensor = None, pool: GearKVCachePool = None, layer: int = 0,
                 heads_total: int = None, tp: dict = None):
        assert head_dim == 128
        cc = compress_config
        self.tp = tp
        v_exact = tp is not None and tp["world"] > 1 and tp.get("exact", True)
        shapes, d = _cache_dims(batch, n_kv_heads, max_tokens, cc, head_dim, heads_total, v_exact)
        self.dims = d
        self.B, self.H, self.D = batch, n_kv_heads, head_dim
        self.bits, self.group, self.R = d["bits"], d["group"], d["R"]
        self.lowrank, self.rk, self.rv = d["lowrank"], d["rk"], d["rv"]
        self.loop = int(cc.get("loop", 3))
        self.fpi, self.Tmax = d["fpi"], d["Tmax"]
        self.kv, self.kk_blk, self.kcap = d["kv"], d["kk_blk"], d["kcap"]
        self.bufs = {}
        for name in ("kcode", "kscale", "kmn", "vcode", "vscale", "vmn", "kPseg", "kQtok", "vPseg", "vQtok", "kwin", "vwin",
                     "koidx", "koval", "voidx", "voval", "kochunk", "vochunk", "ktile", "kcnt", "vtile", "vcnt"):
            if name not in shapes:
                t = None
            elif pool is not None:          # this layer's slice of the pooled storage (contiguous, same layout)
                t = pool.buf[name][layer]
            else:
                shp, dt = shapes[name]
                t = torch.zeros(shp, dtype=dt, device=device)
            setattr(self, name, t)
            if t is not None:
                self.bufs[name] = t
        if pool is not None:
            pool.caches.append(self)
        self.n_comp = 0      # compressed tokens
        self.n_win = 0       # tokens in the fp16 window
        self.seg0 = 0        # tokens of segment 0 (the compressed part of the prompt)
        self.kk0 = 0         # K outliers per side and channel row of segment 0
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(seed)
        self._ws = None
        self.use_tiles = True     # tests switch these off to compare the list paths
        self.use_chunk_index = True
        self._views = {}
        # optional device-side {pos, slot, T, W} shared by all layers (the _dyn methods read it: hipGraph replay)
        self.state = state

    # ------------------------------------------------------------------------------------------------ state
    @property
    def seq_len(self) -> int:
        return self.n_comp + self.n_win

    def _segment_of(self, t0: int) -> int:
        return 0 if t0 < self.seg0 else 1 + (t0 - self.seg0) // self.R

    def _store(self, k_src, v_src, T, seg, kk, o_off):
        assert self.n_comp + T <= self.Tmax, "cache capacity exceeded"
        _compress_into(self.bufs, self.dims, 1, self.B, self.H, self.D, k_src, v_src, T, self.n_comp, seg, kk, o_off, self.loop,
                       self.gen, self.kk0, self.seg0, self.tp)
        self.n_comp += T

    def prefill(self, k: torch.Tensor, v: torch.Tensor):
        """k, v fp16 [B,Hkv,T,128] (post-RoPE): the first T - T % residual tokens are compressed as segment 0, the tail
        stays in the fp16 window (modeling_llamagear.py:390-434)."""
        assert self.seq_len == 0
        T = k.shape[2]
        nq = T - T % self.R
        if nq:
            self.seg0 = nq
            self.kk0 = min(self.dims["kk0_max"], nq // 2) if self.kk_blk else 0
            self._store(k[:, :, :nq].contiguous(), v[:, :, :nq].contiguous(), nq, 0, self.kk0, 0)
        self.n_win = T - nq
        if self.n_win:
            self.kwin[:, :, :self.n_win] = k[:, :, nq:]
            self.vwin[:, :, :self.n_win] = v[:, :, nq:]

    def append_rope(self, qkv: torch.Tensor, n_q_heads: int, pos: int, theta: float) -> torch.Tensor:
        """qkv fp16 [B, (Hq + 2 Hkv) * 128] of the new token -> RoPE, k / v into the window; returns q [B,Hq,1,128]."""
        q = torch.empty((self.B, n_q_heads, 1, self.D), dtype=torch.float16, device=qkv.device)
        rc = L.load().gear_rope_append(L.ptr(qkv), self.B, n_q_heads, self.H, self.D, pos, theta, L.ptr(q), L.ptr(self.kwin),
                                       L.ptr(self.vwin), self.n_win, self.R, L.stream_ptr(qkv))
        L.check(rc, "gear_rope_append")
        self.n_win += 1
        return q

    def append(self, k_new: torch.Tensor, v_new: torch.Tensor):
        """k_new, v_new fp16 [B,Hkv,1,128] (already rotated) into the window."""
        self.kwin[:, :, self.n_win] = k_new[:, :, 0]
        self.vwin[:, :, self.n_win] = v_new[:, :, 0]
        self.n_win += 1

    def _attend(self, q, T, W, dyn):
        B, Hq = q.shape[0], q.shape[1]
        lib = L.load()
        out = torch.empty((B, Hq, 1, self.D), dtype=torch.float16, device=q.device)
        wsb = lib.gear_attn_decode_workspace(B, Hq, self.Tmax, self.bits)
        if self._ws is None or self._ws.numel() < wsb:
            self._ws = torch.empty((wsb,), dtype=torch.uint8, device=q.device)
        key = (self.kk0, self.seg0, bool(W or dyn is not None), self.use_tiles, self.use_chunk_index)
        view = self._views.get(key)
        if view is None:      # the view is a function of the prompt split only: built once per cache, not per token
            view = _view(self.bufs, self.dims, B, self.H, self.D, self.kk0, self.seg0, kwin=key[2])
            if not self.use_tiles:
                view.ktile = view.vtile = None
            if not self.use_chunk_index:
                view.kochunk = view.vochunk = None
            self._views[key] = view
        rc = lib.gear_attn_decode_cache(C_.byref(view), L.ptr(q), Hq, T, W, L.ptr(dyn), 1.0 / math.sqrt(self.D), L.ptr(out), None,
                                        L.ptr(self._ws), self._ws.numel(), L.stream_ptr(q))
        L.check(rc, "gear_attn_decode_cache")
        return out

    def attend(self, q: torch.Tensor) -> torch.Tensor:
        """q fp16 [B,Hq,1,128] -> softmax(q Khat^T / sqrt(128)) Vhat over compressed + window tokens, fp16 [B,Hq,1,128]."""
        return self._attend(q, self.n_comp, self.n_win, None)

    # ---- device-state variants: no host-side counters change here (a captured graph replays these launches) ----------
    def append_rope_dyn(self, qkv: torch.Tensor, n_q_heads: int, theta: float) -> torch.Tensor:
        q = torch.empty((self.B, n_q_heads, 1, self.D), dtype=torch.float16, device=qkv.device)
        rc = L.load().gear_rope_append_dyn(L.ptr(qkv), self.B, n_q_heads, self.H, self.D, L.ptr(self.state), theta, L.ptr(q),
                                           L.ptr(self.kwin), L.ptr(self.vwin), self.R, L.stream_ptr(qkv))
        L.check(rc, "gear_rope_append_dyn")
        return q

    def attend_dyn(self, q: torch.Tensor) -> torch.Tensor:
        return self._attend(q, self.Tmax, self.R, self.state)

    def maybe_compress(self):
        """Compress the window when it holds `residual` tokens (modeling_llamagear.py:265, :335)."""
        if self.n_win == self.R:
            seg = self._segment_of(self.n_comp)
            o_off = self.kk0 + ((self.n_comp - self.seg0) // self.R) * self.kk_blk
            self._store(self.kwin, self.vwin, self.R, seg, self.kk_blk, o_off)
            self.n_win = 0


class Fp16KVCache:
    """The UNCOMPRESSED baseline behind the same decoder (the model "None" the reference's harness times beside gearl / KIVI,
    cuda_supported_gear/test.py:41-62): every token's K / V stays fp16 in a pre-allocated [B, Hkv, capacity, 128] pair, the
    attention is gear_attn_decode_f16 (the same split / merge kernels as the compressed path).  It has GearKVCache's window
    interface -- the "window" is the whole cache --, so FastGearDecoder's fused q/k/v projection appends to it unchanged."""

    MAX_TOKENS = 8320      # gear_attn_decode_f16: 65 chunks of 128 tokens in one merge (csrc/attention.hip RS_MAX)

    def __init__(self, batch: int, n_kv_heads: int, max_tokens: int, device, head_dim: int = 128):
        if max_tokens > self.MAX_TOKENS:
            raise L.GearError(f"Fp16KVCache: capacity {max_tokens} > {self.MAX_TOKENS} tokens, the longest context "
                              "gear_attn_decode_f16 merges in one launch (the baseline would fail at its first attend())")
        self.B, self.H, self.D = batch, n_kv_heads, head_dim
        self.R = max_tokens
        self.kwin = torch.zeros((batch, n_kv_heads, max_tokens, head_dim), dtype=torch.float16, device=device)
        self.vwin = torch.zeros_like(self.kwin)
        self.n_win = 0
        self.n_comp = 0
        self.state = None

    @property
    def seq_len(self) -> int:
        return self.n_win

    def prefill(self, k: torch.Tensor, v: torch.Tensor):
        assert self.n_win == 0 and k.shape[2] <= self.R
        T = k.shape[2]
        self.kwin[:, :, :T] = k
        self.vwin[:, :, :T] = v
        self.n_win = T

    def append_rope(self, qkv: torch.Tensor, n_q_heads: int, pos: int, theta: float) -> torch.Tensor:
        if self.n_win >= self.R:
            raise L.GearError(f"Fp16KVCache: capacity {self.R} reached")
        q = torch.empty((self.B, n_q_heads, 1, self.D), dtype=torch.float16, device=qkv.device)
        rc = L.load().gear_rope_append(L.ptr(qkv), self.B, n_q_heads, self.H, self.D, pos, theta, L.ptr(q), L.ptr(self.kwin),
                                       L.ptr(self.vwin), self.n_win, self.R, L.stream_ptr(qkv))
        L.check(rc, "gear_rope_append")
        self.n_win += 1
        return q

    def attend(self, q: torch.Tensor) -> torch.Tensor:
        from .attention import decode_attention_f16
        return decode_attention_f16(q, self.kwin, self.vwin, T=self.n_win)

