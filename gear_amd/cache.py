"""Pre-allocated, kernel-native GEAR KV cache for one attention layer (the build's own
design: the reference rebuilds every payload tensor with torch.cat on each block boundary,
cuda_supported_gear/modeling_llamagear.py:273-286, :365-378, and re-lays the whole packed K out on every token,
quant/matmul.py:205, :215-216).

State machine = the attention hook's (modeling_llamagear.py:177-484): an fp16 window of the most recent < `residual`
tokens; when it fills, the block is compressed (quantize + pack + per-block rank-r factors) in place behind the already
compressed tokens.  Layout = what gear_attn_decode_seg streams: K channel-major with a fixed row pitch (so a block append
is 128 short row segments, never a re-layout), V token-major, token-side factors per token, channel-side factors per
segment (segment 0 = the prompt, then one per block).

Differences from the hook, on purpose: K and V are compressed in lockstep (the hook compresses V only when T > residual,
:416, which strands V in fp16 when the prompt is exactly `residual` long), and the block factors start from a
channel-side random basis (the simulated path's orientation, compress_function.py:83) so that the Gram-matrix kernel
applies; both are rank-r power-iteration approximations of the same error matrix.
"""
from __future__ import annotations

import math

import torch

from . import _lib as L
from . import compress as C


def _cache_dims(batch, n_kv_heads, max_tokens, cc, head_dim=128):
    bits, group, R = cc["quantize_bit"], cc["group_size"], cc["residual"]
    m = cc["compress_method"]
    lowrank = ("gearl" in m) or ("gearsl" in m)
    rk = int(cc["rank"]) if lowrank else 0
    rv = int(cc["rankv"]) if lowrank else 0
    assert R % 64 == 0 and R % group == 0
    fpi = 32 // bits
    Tmax = (max_tokens + R - 1) // R * R
    nseg = 1 + Tmax // R
    B, H, D, T = batch, n_kv_heads, head_dim, Tmax
    shapes = dict(kcode=((B, H, D, T // fpi), torch.int32), kscale=((B, H, D, T // group), torch.float16),
                  kmn=((B, H, D, T // group), torch.float16), vcode=((B, H, T, D // fpi), torch.int32),
                  vscale=((B, H, T, D // group), torch.float16), vmn=((B, H, T, D // group), torch.float16),
                  kwin=((B, H, R, D), torch.float16), vwin=((B, H, R, D), torch.float16))
    if lowrank:
        shapes.update(kPseg=((nseg, B, H, D, rk), torch.float16), kQtok=((B, H, T, rk), torch.float16),
                      vPseg=((nseg, B, H, D, rv), torch.float16), vQtok=((B, H, T, rv), torch.float16))
    return shapes, dict(bits=bits, group=group, R=R, lowrank=lowrank, rk=rk, rv=rv, fpi=fpi, Tmax=Tmax, nseg=nseg)


class GearKVCachePool:
    """The buffers of ALL layers' caches as one tensor per field with a leading layer dimension.  Every layer's
    GearKVCache takes its (contiguous) slice, so nothing changes for the kernels; what the pool buys is the block boundary:
    all layers fill their fp16 windows on the same token, and with pooled storage the 32 per-layer compress + 10-copy
    sequences (~800 launches) become ONE compress_key / compress_value over [layers * batch, H, residual, 128] and one
    strided copy per field (compress_all)."""

    def __init__(self, n_layers: int, batch: int, n_kv_heads: int, max_tokens: int, compress_config: dict, device,
                 head_dim: int = 128, seed: int = 0):
        shapes, self.dims = _cache_dims(batch, n_kv_heads, max_tokens, compress_config, head_dim)
        self.L, self.B, self.H, self.D = n_layers, batch, n_kv_heads, head_dim
        self.loop = int(compress_config.get("loop", 3))
        self.buf = {n: torch.zeros((n_layers,) + shp, dtype=dt, device=device) for n, (shp, dt) in shapes.items()}
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(seed)
        self.caches = []

    def compress_all(self):
        """Every layer's window holds `residual` tokens: compress all of them in one go and append behind the compressed
        tokens (same arithmetic as GearKVCache._store_block, the layers ride in the batch dimension)."""
        d, b, c0 = self.dims, self.buf, self.caches[0]
        L_, B, H, D, R, g, fpi = self.L, self.B, self.H, self.D, d["R"], d["group"], d["fpi"]
        assert all(c.n_win == R and c.n_comp == c0.n_comp for c in self.caches)
        t0, seg = c0.n_comp, c0._segment_of(c0.n_comp)
        assert t0 + R <= d["Tmax"], "cache capacity exceeded"
        P0k = P0v = None
        if d["lowrank"]:
            P0k = torch.rand((L_ * B, H, D, d["rk"]), device=b["kwin"].device, generator=self.gen)
            P0v = torch.rand((L_ * B, H, D, d["rv"]), device=b["kwin"].device, generator=self.gen)
        pk = C.compress_key(b["kwin"].view(L_ * B, H, R, D), d["bits"], g, rank=d["rk"], loop=self.loop, mode="fp16", P0=P0k)
        pv = C.compress_value(b["vwin"].view(L_ * B, H, R, D), d["bits"], g, rank=d["rv"], loop=self.loop, mode="fp16", P0=P0v)
        b["kcode"][..., t0 // fpi:(t0 + R) // fpi] = pk.code.view(L_, B, H, D, R // fpi)
        b["kscale"][..., t0 // g:(t0 + R) // g] = pk.scale.view(L_, B, H, D, R // g)
        b["kmn"][..., t0 // g:(t0 + R) // g] = pk.mn.view(L_, B, H, D, R // g)
        b["vcode"][:, :, :, t0:t0 + R] = pv.code.view(L_, B, H, R, D // fpi)
        b["vscale"][:, :, :, t0:t0 + R] = pv.scale.view(L_, B, H, R, D // g)
        b["vmn"][:, :, :, t0:t0 + R] = pv.mn.view(L_, B, H, R, D // g)
        if d["lowrank"]:
            b["kPseg"][:, seg] = pk.P.view(L_, B, H, D, d["rk"])
            b["kQtok"][:, :, :, t0:t0 + R] = pk.Q.view(L_, B, H, R, d["rk"])
            b["vPseg"][:, seg] = pv.P.view(L_, B, H, D, d["rv"])
            b["vQtok"][:, :, :, t0:t0 + R] = pv.Q.view(L_, B, H, R, d["rv"])
        for c in self.caches:
            c.n_comp += R
            c.n_win = 0


class GearKVCache:
    def __init__(self, batch: int, n_kv_heads: int, max_tokens: int, compress_config: dict, device, head_dim: int = 128,
                 seed: int = 0, state: torch.Tensor = None, pool: GearKVCachePool = None, layer: int = 0):
        assert head_dim == 128
        cc = compress_config
        shapes, d = _cache_dims(batch, n_kv_heads, max_tokens, cc, head_dim)
        self.B, self.H, self.D = batch, n_kv_heads, head_dim
        self.bits, self.group, self.R = d["bits"], d["group"], d["R"]
        self.lowrank, self.rk, self.rv = d["lowrank"], d["rk"], d["rv"]
        self.loop = int(cc.get("loop", 3))
        self.fpi, self.Tmax = d["fpi"], d["Tmax"]
        for name in ("kcode", "kscale", "kmn", "vcode", "vscale", "vmn", "kPseg", "kQtok", "vPseg", "vQtok", "kwin", "vwin"):
            if name not in shapes:
                setattr(self, name, None)
            elif pool is not None:          # this layer's slice of the pooled storage (contiguous, same layout)
                setattr(self, name, pool.buf[name][layer])
            else:
                shp, dt = shapes[name]
                setattr(self, name, torch.zeros(shp, dtype=dt, device=device))
        if pool is not None:
            pool.caches.append(self)
        self.n_comp = 0      # compressed tokens
        self.n_win = 0       # tokens in the fp16 window
        self.seg0 = 0        # tokens of segment 0 (the compressed part of the prompt)
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(seed)
        self._ws = None
        # optional device-side {pos, slot, T, W} shared by all layers (the _dyn methods read it: hipGraph replay)
        self.state = state

    # ------------------------------------------------------------------------------------------------ state
    @property
    def seq_len(self) -> int:
        return self.n_comp + self.n_win

    def _segment_of(self, t0: int) -> int:
        return 0 if t0 < self.seg0 else 1 + (t0 - self.seg0) // self.R

    def _store_block(self, k_blk: torch.Tensor, v_blk: torch.Tensor, seg: int):
        """Compress [B,H,n,128] K and V and write them behind the compressed tokens."""
        n, t0, g, fpi = k_blk.shape[2], self.n_comp, self.group, self.fpi
        assert t0 + n <= self.Tmax, "cache capacity exceeded"
        P0k = P0v = None
        if self.lowrank:
            P0k = torch.rand((self.B, self.H, self.D, self.rk), device=k_blk.device, generator=self.gen)
            P0v = torch.rand((self.B, self.H, self.D, self.rv), device=k_blk.device, generator=self.gen)
        pk = C.compress_key(k_blk, self.bits, g, rank=self.rk, loop=self.loop, mode="fp16", P0=P0k)
        pv = C.compress_value(v_blk, self.bits, g, rank=self.rv, loop=self.loop, mode="fp16", P0=P0v)
        self.kcode[:, :, :, t0 // fpi:(t0 + n) // fpi] = pk.code
        self.kscale[:, :, :, t0 // g:(t0 + n) // g] = pk.scale
        self.kmn[:, :, :, t0 // g:(t0 + n) // g] = pk.mn
        self.vcode[:, :, t0:t0 + n] = pv.code
        self.vscale[:, :, t0:t0 + n] = pv.scale
        self.vmn[:, :, t0:t0 + n] = pv.mn
        if self.lowrank:
            self.kPseg[seg] = pk.P
            self.kQtok[:, :, t0:t0 + n] = pk.Q
            self.vPseg[seg] = pv.P
            self.vQtok[:, :, t0:t0 + n] = pv.Q
        self.n_comp += n

    def prefill(self, k: torch.Tensor, v: torch.Tensor):
        """k, v fp16 [B,Hkv,T,128] (post-RoPE): the first T - T % residual tokens are compressed as segment 0, the tail
        stays in the fp16 window (modeling_llamagear.py:390-434)."""
        assert self.seq_len == 0
        T = k.shape[2]
        nq = T - T % self.R
        if nq:
            self.seg0 = nq
            self._store_block(k[:, :, :nq].contiguous(), v[:, :, :nq].contiguous(), 0)
        self.n_win = T - nq
        if self.n_win:
            self.kwin[:, :, :self.n_win] = k[:, :, nq:]
            self.vwin[:, :, :self.n_win] = v[:, :, nq:]

    def append_rope(self, qkv: torch.Tensor, n_q_heads: int, pos: int, theta: float) -> torch.Tensor:
        """qkv fp16 [B, (Hq + 2 Hkv) * 128] of the new token -> RoPE, k / v into the window; returns q [B,Hq,1,128]."""
        q = torch.empty((self.B, n_q_heads, 1, self.D), dtype=torch.float16, device=qkv.device)
        rc = L.load().gear_rope_append(L.ptr(qkv), self.B, n_q_heads, self.H, self.D, pos, theta, L.ptr(q), L.ptr(self.kwin),
                                       L.ptr(self.vwin), self.n_win, self.R, L.stream_ptr())
        L.check(rc, "gear_rope_append")
        self.n_win += 1
        return q

    def append(self, k_new: torch.Tensor, v_new: torch.Tensor):
        """k_new, v_new fp16 [B,Hkv,1,128] (already rotated) into the window."""
        self.kwin[:, :, self.n_win] = k_new[:, :, 0]
        self.vwin[:, :, self.n_win] = v_new[:, :, 0]
        self.n_win += 1

    def attend(self, q: torch.Tensor) -> torch.Tensor:
        """q fp16 [B,Hq,1,128] -> softmax(q Khat^T / sqrt(128)) Vhat over compressed + window tokens, fp16 [B,Hq,1,128]."""
        B, Hq = q.shape[0], q.shape[1]
        lib = L.load()
        out = torch.empty((B, Hq, 1, self.D), dtype=torch.float16, device=q.device)
        wsb = lib.gear_attn_decode_workspace(B, Hq, self.Tmax, self.bits)
        if self._ws is None or self._ws.numel() < wsb:
            self._ws = torch.empty((wsb,), dtype=torch.uint8, device=q.device)
        p = L.ptr
        T = self.n_comp
        rc = lib.gear_attn_decode_seg(
            p(q), p(self.kcode), p(self.kscale), p(self.kmn), p(self.kPseg), p(self.kQtok), None, None,
            p(self.vcode), p(self.vscale), p(self.vmn), p(self.vPseg), p(self.vQtok), None, None,
            p(self.kwin) if self.n_win else None, p(self.vwin) if self.n_win else None,
            B, Hq, self.H, self.D, T, self.n_win, self.Tmax // self.fpi, self.Tmax // self.group, self.Tmax, self.Tmax,
            self.Tmax, self.group, self.bits, 0, self.rk, self.rv, 0, 0, self.seg0, self.R if self.lowrank else 0, self.R,
            1.0 / math.sqrt(self.D), p(out), None, p(self._ws), self._ws.numel(), L.stream_ptr())
        L.check(rc, "gear_attn_decode_seg")
        return out

    # ---- device-state variants: no host-side counters change here (a captured graph replays these launches) ----------
    def append_rope_dyn(self, qkv: torch.Tensor, n_q_heads: int, theta: float) -> torch.Tensor:
        q = torch.empty((self.B, n_q_heads, 1, self.D), dtype=torch.float16, device=qkv.device)
        rc = L.load().gear_rope_append_dyn(L.ptr(qkv), self.B, n_q_heads, self.H, self.D, L.ptr(self.state), theta, L.ptr(q),
                                           L.ptr(self.kwin), L.ptr(self.vwin), self.R, L.stream_ptr())
        L.check(rc, "gear_rope_append_dyn")
        return q

    def attend_dyn(self, q: torch.Tensor) -> torch.Tensor:
        B, Hq = q.shape[0], q.shape[1]
        lib = L.load()
        out = torch.empty((B, Hq, 1, self.D), dtype=torch.float16, device=q.device)
        wsb = lib.gear_attn_decode_workspace(B, Hq, self.Tmax, self.bits)
        if self._ws is None or self._ws.numel() < wsb:
            self._ws = torch.empty((wsb,), dtype=torch.uint8, device=q.device)
        p = L.ptr
        rc = lib.gear_attn_decode_dyn(
            p(q), p(self.kcode), p(self.kscale), p(self.kmn), p(self.kPseg), p(self.kQtok), None, None,
            p(self.vcode), p(self.vscale), p(self.vmn), p(self.vPseg), p(self.vQtok), None, None, p(self.kwin), p(self.vwin),
            B, Hq, self.H, self.D, self.Tmax, self.R, self.Tmax // self.fpi, self.Tmax // self.group, self.Tmax, self.Tmax,
            self.Tmax, self.group, self.bits, 0, self.rk, self.rv, 0, 0, self.seg0, self.R if self.lowrank else 0, self.R,
            p(self.state), 1.0 / math.sqrt(self.D), p(out), None, p(self._ws), self._ws.numel(), L.stream_ptr())
        L.check(rc, "gear_attn_decode_dyn")
        return out

    def maybe_compress(self):
        """Compress the window when it holds `residual` tokens (modeling_llamagear.py:265, :335)."""
        if self.n_win == self.R:
            self._store_block(self.kwin, self.vwin, self._segment_of(self.n_comp))
            self.n_win = 0
