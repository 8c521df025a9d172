"""Counterpart of cuda_supported_gear/modeling_llamagear.py: the attention hook that owns the GEAR KV cache.

Same operator surface as the reference --
    key_compression(key_full, compress_config)            modeling_llamagear.py:23-37
    value_compression(value_full, compress_config)        :39-53
    matmul_withlrap(group_size, a, b, scale, mn, bits, pbase, qbase, type)   :54-111
    LlamaAttention_GEAR(layer_idx, config, compress_config).forward(...) -> (attn_output, None, 17-tuple)   :113-484
    LlamaForCausalLM_GEARKIVI(config, compress_config)    :711
-- on the HIP kernels of libgear_hip.so.  The module is self-contained (no `transformers` import: the reference is
a fork of transformers 4.38 that no longer imports under the installed 5.x); `config` may be a HF LlamaConfig or
the LlamaConfigLite below (same attribute names).

Cache tuple (one per layer), identical slot layout to the reference (:458-466):
  0 K code  int32 [B,Hkv,D,Tq/fpi]   1 K_full fp16 [B,Hkv,t<R,D] | None   2 K scale [B,Hkv,D,Tq/g]   3 K mn
  4 V code  int32 [B,Hkv,Tq,D/fpi]   5 V_full fp16 | None                 6 V scale [B,Hkv,Tq,D/g]   7 V mn
  8 kv_seq_len (int)                 9 K P list   10 K Q list   11, 12 None (K outliers: not stored by the
  reference's fused path)            13 V P list  14 V Q list   15, 16 None
  P/Q lists: [prefill factors] or [prefill factors, stacked decode-block factors [nbuf,B,Hkv,.,r]].
  K: P is the token-side factor [.,T,r], Q the channel-side factor [.,D,r]; V: P [.,D,r], Q [.,T,r] (:227-230).

Documented divergences from the reference: B1 (all code columns are packed), B2 (the K
low-rank factors approximate the true error matrix, not a reshape-scrambled one), B3 (per-batch bases), GQA is
supported (the reference asserts num_key_value_groups == 1, :206).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib as L
from .quant.matmul import cuda_bmm_fA_qB_outer
from .quant.new_pack import (headwise_lrap, triton_quantize_and_pack_along_last_dim,
                             triton_quantize_and_pack_along_last_dim_witherror)


@dataclass
class LlamaConfigLite:
    """The LlamaConfig attributes the GEAR attention reads (defaults: Llama-2-7B; CSG/test.py:12-17 adds the
    k_bits / v_bits / group_size / residual_length fields)."""
    vocab_size: int = 32000
    hidden_size: int = 4096
    intermediate_size: int = 11008
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    num_key_value_heads: int = 32
    max_position_embeddings: int = 4096
    rms_norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    attention_bias: bool = False
    attention_dropout: float = 0.0
    pretraining_tp: int = 1
    rope_scaling: Optional[dict] = None
    k_bits: int = 2
    v_bits: int = 2
    group_size: int = 64
    residual_length: int = 64


def _uses_lowrank(compress_config) -> bool:
    m = compress_config["compress_method"]
    return "gearl" in m or "gearsl" in m          # substring test, modeling_llamagear.py:25 / :41


def key_compression(key_full: torch.Tensor, compress_config: dict):
    """modeling_llamagear.py:23-37.  key_full fp16 [B,H,D,T] (K^T, contiguous) ->
    (code [B,H,D,T/fpi], scale [B,H,D,T/g], mn, P [B,H,T,r] | None, Q [B,H,D,r] | None)."""
    bsz, num_head, head_dim, seq_len = key_full.shape
    if _uses_lowrank(compress_config):
        code, scale, mn, error = triton_quantize_and_pack_along_last_dim_witherror(
            key_full, compress_config["group_size"], compress_config["quantize_bit"])
        # B2: the error tensor is E^T [B,H,D,T]; the reference reshapes it as [B,H,T,D] and transposes (a scramble)
        error = error.view(bsz, num_head, head_dim, seq_len)
        key_states_p, key_states_q = headwise_lrap(error, compress_config["rank"], compress_config["loop"])
    else:
        code, scale, mn = triton_quantize_and_pack_along_last_dim(
            key_full, compress_config["group_size"], compress_config["quantize_bit"])
        key_states_p, key_states_q = None, None
    return code, scale, mn, key_states_p, key_states_q


def value_compression(value_full: torch.Tensor, compress_config: dict):
    """modeling_llamagear.py:39-53.  value_full fp16 [B,H,T,D] ->
    (code [B,H,T,D/fpi], scale [B,H,T,D/g], mn, P [B,H,D,r] | None, Q [B,H,T,r] | None)."""
    bsz, num_head, seq_len, head_dim = value_full.shape
    if _uses_lowrank(compress_config):
        code, scale, mn, error = triton_quantize_and_pack_along_last_dim_witherror(
            value_full, compress_config["group_size"], compress_config["quantize_bit"])
        error = error.view(bsz, num_head, seq_len, head_dim)
        value_states_p, value_states_q = headwise_lrap(error, compress_config["rankv"], compress_config["loop"])
    else:
        code, scale, mn = triton_quantize_and_pack_along_last_dim(
            value_full, compress_config["group_size"], compress_config["quantize_bit"])
        value_states_p, value_states_q = None, None
    return code, scale, mn, value_states_p, value_states_q


def _rep(t: torch.Tensor, n_rep: int) -> torch.Tensor:
    """[.., B, Hkv, x, y] -> [.., B, Hkv*n_rep, x, y] (GQA: query heads of one KV head are adjacent)."""
    return t if n_rep == 1 else t.repeat_interleave(n_rep, dim=-3)


def matmul_withlrap(group_size, a, b, scale, mn, bits, pbase: list, qbase: list, type="key"):
    """modeling_llamagear.py:54-111: dequant GEMV + low-rank correction -- here ONE call into the HIP library
    (gear_gemv_outer_lrap: the GEMV kernel writes fp32 partial sums, its split-K epilogue adds the factor terms and rounds once)
    where the reference adds the correction with ~8 eager matmul / permute / slice-assign launches.

    a [B,Hq,1,K] fp16; b packed [B,Hkv,K,N/fpi]; type "key": K = head_dim, N = compressed tokens; type "value": K = compressed
    tokens, N = head_dim.  pbase / qbase: [None] | [prefill] | [prefill, stacked [nbuf,B,Hkv,.,r]] (:71-85 / :87-108):
    key P [B,Hkv,T,r] (token side), Q [B,Hkv,D,r]; value P [B,Hkv,D,r], Q [B,Hkv,T,r]."""
    if pbase[0] is None:
        return cuda_bmm_fA_qB_outer(group_size, a, b, scale, mn, bits)
    assert type in ("key", "value") and bits in (2, 4) and a.dim() == 4 and b.dim() == 4
    B, Hq, M, K = a.shape
    if M != 1:
        raise L.GearError("matmul_withlrap supports q_len == 1 only (decode GEMV)")
    Hkv = b.shape[1]
    N = b.shape[-1] * (32 // bits)
    p0, q0 = pbase[0].contiguous(), qbase[0].contiguous()
    r = p0.shape[-1]
    tp = (p0 if type == "key" else q0).shape[-2]
    p1 = q1 = None
    nbuf, blk = 0, 64
    if len(pbase) > 1:
        p1, q1 = pbase[1].contiguous(), qbase[1].contiguous()
        nbuf, blk = p1.shape[0], (p1 if type == "key" else q1).shape[-2]
    a, b, scale, mn = a.contiguous(), b.contiguous(), scale.contiguous(), mn.contiguous()
    L.require_gpu(a, b, scale, mn, p0, q0, p1, q1)
    if any(t is not None and t.dtype != torch.float16 for t in (a, p0, q0, p1, q1)):
        raise L.GearError("matmul_withlrap: activations and factors must be float16")
    # the epilogue kernel's geometry (csrc/gemv.hip: blocks of 16 .. 64 tokens, rank <= 16, head_dim <= 256 / <= 128 columns);
    # anything else -- e.g. residual = 128 blocks -- keeps the reference's shape: the HIP dequant GEMV + the factor terms as
    # batched fp16 matmuls on the GPU (modeling_llamagear.py:71-108)
    if not (16 <= blk <= 64 and r <= 16 and (K <= 256 if type == "key" else N <= 128)):
        return _matmul_withlrap_general(group_size, a, b, scale, mn, bits, p0, q0, p1, q1, tp, nbuf, blk, type, Hq // Hkv)
    lib = L.load()
    BA = B * Hq
    wsb = lib.gear_gemv_outer_lrap_workspace(BA, K, N, bits)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=a.device)
    out = torch.empty((B, Hq, 1, N), dtype=torch.float16, device=a.device)
    rc = lib.gear_gemv_outer_lrap(L.ptr(a), L.ptr(b), L.ptr(scale), L.ptr(mn), BA, Hq // Hkv, K, N, group_size, bits,
                                  0 if scale.dtype == torch.float16 else 1, 0 if type == "key" else 1, L.ptr(p0), L.ptr(q0), tp,
                                  L.ptr(p1), L.ptr(q1), nbuf, blk, r, L.ptr(out), L.ptr(ws), wsb, L.stream_ptr(a))
    L.check(rc, "gear_gemv_outer_lrap")
    return out


def _matmul_withlrap_general(group_size, a, b, scale, mn, bits, p0, q0, p1, q1, tp, nbuf, blk, type, n_rep):
    """matmul_withlrap for factor geometries outside the fused epilogue's limits: the dequant GEMV stays the HIP kernel, the
    low-rank terms are the reference's own batched matmuls (modeling_llamagear.py:71-85 key, :87-108 value), GQA by repeating
    the KV heads' factors."""
    out = cuda_bmm_fA_qB_outer(group_size, a, b, scale, mn, bits)
    if type == "key":
        out[..., :tp] += (a @ _rep(q0, n_rep)) @ _rep(p0, n_rep).transpose(-1, -2)
        if nbuf:
            t = (a.unsqueeze(0) @ _rep(q1, n_rep)) @ _rep(p1, n_rep).transpose(-1, -2)       # [nbuf,B,Hq,1,blk]
            out[..., tp:tp + nbuf * blk] += t.permute(1, 2, 3, 0, 4).reshape(*out.shape[:3], nbuf * blk)
    else:
        out += (a[..., :tp] @ _rep(q0, n_rep)) @ _rep(p0, n_rep).transpose(-1, -2)
        if nbuf:
            B, Hq = a.shape[:2]
            ab = a[..., tp:tp + nbuf * blk].reshape(B, Hq, 1, nbuf, blk).permute(3, 0, 1, 2, 4)  # [nbuf,B,Hq,1,blk]
            out += ((ab @ _rep(q1, n_rep)) @ _rep(p1, n_rep).transpose(-1, -2)).sum(0)
    return out


# ------------------------------------------------------------------------------------------------- RoPE (Llama)
class LlamaRotaryEmbedding(nn.Module):
    def __init__(self, dim, max_position_embeddings=2048, base=10000.0):
        super().__init__()
        self.dim, self.base = dim, base
        inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim))
        self.register_buffer("inv_freq", inv_freq, persistent=False)

    @torch.no_grad()
    def forward(self, x, position_ids):
        freqs = position_ids[:, :, None].float() * self.inv_freq[None, None, :].to(x.device)
        emb = torch.cat((freqs, freqs), dim=-1)
        return emb.cos().to(x.dtype), emb.sin().to(x.dtype)


def rotate_half(x):
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def apply_rotary_pos_emb(q, k, cos, sin):
    cos, sin = cos.unsqueeze(1), sin.unsqueeze(1)
    return (q * cos) + (rotate_half(q) * sin), (k * cos) + (rotate_half(k) * sin)


def _append(old, new, dim):
    return new if old is None else torch.cat([old, new], dim=dim)


def _append_factor(lst: list, new: torch.Tensor):
    """Decode-block factors are stacked on a leading buffer dim in slot [1] (modeling_llamagear.py:276-286)."""
    new = new.unsqueeze(0)
    if len(lst) == 1:
        lst.append(new)
    else:
        lst[1] = torch.cat([lst[1], new], dim=0)


class GearHookCache:
    """The 17-slot cache tuple of LlamaAttention_GEAR (modeling_llamagear.py:458-466) as a VIEW of a pre-allocated, kernel-native
    GearKVCache: slot 8 (kv_seq_len, the one slot the model reads, :624 / :838) is a plain int, every other slot is materialised
    when somebody indexes it -- slices of the cache tensors with the lengths this step had, the factor lists re-shaped into the
    reference's [prefill, stacked [nbuf,B,H,.,r]] form.  The attention hook recognises the object and takes its fast decode path
    (window append in place, ONE fused attention call over the packed cache, compress of a full window in place): no torch.cat of
    the payload, no per-token GEMV pair + eager softmax.  len() == 17, iteration and indexing behave like the tuple.

    SINGLE USE.  Unlike the reference's tuple (immutable: an earlier `past` can be fed again, rolled back to, or branched from, as
    assisted / beam decoding do), the object is a view of ONE cache that every decode step mutates in place.  A decode step
    consumes the object it was given and returns a new one; feeding a consumed one (or any older one) to a decode step again
    raises GearError instead of silently attending over an extra token at a wrong position.  Reading a stale object stays
    possible where it is still TRUE: the packed slots are prefixes of buffers that only grow (compressed tokens never change), the
    window slots 1 / 5 are readable until a block boundary has overwritten the window -- after that they raise too.  A caller
    that needs to branch calls materialize() BEFORE the step that consumes the object and continues on the tuple."""

    def __init__(self, cache, lowrank: bool, seq_len: int):
        self.cache, self.lowrank = cache, lowrank
        self.n_comp, self.n_win, self.seg0, self.seq_len = cache.n_comp, cache.n_win, cache.seg0, seq_len
        # generation of the shared cache this view belongs to: every mutation (append / block compress) moves the cache on
        self.gen = cache.hook_gen = getattr(cache, "hook_gen", 0) + 1

    def check_live(self):
        """Raise unless this is the newest view of its cache (see the class docstring): what a decode step requires."""
        c = self.cache
        if getattr(c, "hook_gen", self.gen) != self.gen or (c.n_comp, c.n_win) != (self.n_comp, self.n_win):
            raise L.GearError("GearHookCache: this past_key_value was already consumed by a decode step (the cache behind it is "
                              "mutated in place); re-using, rolling back to or branching from an earlier past needs "
                              "materialize() before the step that consumes it")

    def _window_intact(self):
        """The fp16 window as this view saw it is still in the buffer: no block boundary since (same compressed length) and the
        live window is at least as long."""
        c = self.cache
        if c.n_comp != self.n_comp or c.n_win < self.n_win:
            raise L.GearError("GearHookCache: the fp16 window of this (stale) past_key_value has been overwritten by a block "
                              "boundary; materialize() before the step that consumes the object keeps it")

    def __len__(self):
        return 17

    def __iter__(self):
        return (self[i] for i in range(17))

    def _factors(self, tok, seg):
        """token-side [B,H,T,r] + per-segment channel-side [nseg,B,H,D,r] -> (token list, channel list) in the reference's form."""
        c = self.cache
        if not self.lowrank or self.n_comp == 0:
            return [None], [None]
        R = c.R
        if self.seg0:
            first_t, first_c, t1, s1 = tok[:, :, :self.seg0], seg[0], self.seg0, 1
        else:
            first_t, first_c, t1, s1 = tok[:, :, :R], seg[1], R, 2
        tl, cl = [first_t], [first_c]
        nb = (self.n_comp - t1) // R
        if nb > 0:
            B, H, r = tok.shape[0], tok.shape[1], tok.shape[-1]
            tl.append(tok[:, :, t1:self.n_comp].reshape(B, H, nb, R, r).permute(2, 0, 1, 3, 4))
            cl.append(seg[s1:s1 + nb])
        return tl, cl

    def __getitem__(self, i):
        if isinstance(i, slice):
            return tuple(self[j] for j in range(*i.indices(17)))
        if i < 0:
            i += 17
        c, n, w = self.cache, self.n_comp, self.n_win
        if i == 8:
            return self.seq_len
        if i in (11, 12, 15, 16):
            return None
        if i in (1, 5):
            if w == 0:
                return None
            self._window_intact()
            return (c.kwin if i == 1 else c.vwin)[:, :, :w]
        if i in (0, 2, 3):
            if n == 0:
                return None
            t, div = ((c.kcode, c.fpi), (c.kscale, c.group), (c.kmn, c.group))[(0, 2, 3).index(i)]
            return t[..., :n // div]
        if i in (4, 6, 7):
            return None if n == 0 else (c.vcode, c.vscale, c.vmn)[(4, 6, 7).index(i)][:, :, :n]
        if i in (9, 10):
            tl, cl = self._factors(c.kQtok, c.kPseg) if self.lowrank else ([None], [None])
            return tl if i == 9 else cl                   # K: P = token side (:227-230), Q = channel side
        if i in (13, 14):
            tl, cl = self._factors(c.vQtok, c.vPseg) if self.lowrank else ([None], [None])
            return cl if i == 13 else tl                  # V: P = channel side, Q = token side
        raise IndexError(i)

    def materialize(self) -> tuple:
        """The plain tuple (own storage): what the tuple-based path of the hook continues from."""
        def own(x):
            if isinstance(x, list):
                return [own(y) for y in x]
            return x.contiguous().clone() if torch.is_tensor(x) else x
        return tuple(own(self[i]) for i in range(17))


_cache_full_warned = False


def _warn_cache_full(c):
    """The pre-allocated cache behind a GearHookCache is full: the hook continues on the reference-shaped tuple path (materialize(),
    torch.cat appends, ~60 eager launches per layer and token) -- correct, several times slower.  Said once per process."""
    global _cache_full_warned
    if not _cache_full_warned:
        _cache_full_warned = True
        import warnings
        warnings.warn(f"gear_amd: the pre-allocated KV cache ({c.Tmax} tokens = prompt + config.gear_max_new_tokens) is full; decoding "
                      "continues on the tuple-shaped path, which is several times slower -- set config.gear_max_new_tokens to the "
                      "number of tokens you generate", RuntimeWarning, stacklevel=3)


class LlamaAttention_GEAR(nn.Module):
    """modeling_llamagear.py:113-484: attention whose cache is the packed GEAR payload plus an fp16 residual
    window of `residual` tokens; a block is compressed whenever the window fills."""

    def __init__(self, layer_idx, config, compress_config=None, tp_rank: int = 0, tp_world: int = 1, tp_group=None):
        """tp_world > 1: head-sharded attention (new design -- the reference has no
        distributed code): this rank owns num_attention_heads / tp_world query heads and the matching KV heads, its
        own slice of the packed cache, and all-gathers the per-head attention output before the (replicated) o_proj."""
        super().__init__()
        self.layer_idx = layer_idx
        self.compress_config = compress_config
        self.config = config
        self.tp_rank, self.tp_world, self.tp_group = tp_rank, tp_world, tp_group
        self.hidden_size = config.hidden_size
        self.total_heads = config.num_attention_heads
        self.head_dim = self.hidden_size // self.total_heads
        if config.num_attention_heads % tp_world or config.num_key_value_heads % tp_world:
            raise ValueError(f"heads ({config.num_attention_heads} q / {config.num_key_value_heads} kv) must divide "
                             f"across {tp_world} ranks")
        self.num_heads = config.num_attention_heads // tp_world              # local query heads
        self.num_key_value_heads = config.num_key_value_heads // tp_world    # local KV heads
        self.num_key_value_groups = self.num_heads // self.num_key_value_heads
        self.max_position_embeddings = config.max_position_embeddings
        self.rope_theta = config.rope_theta
        self.k_bits = config.k_bits
        self.v_bits = config.v_bits
        self.group_size = config.group_size
        self.residual_length = compress_config["residual"]
        if (self.head_dim * self.total_heads) != self.hidden_size:
            raise ValueError(f"hidden_size must be divisible by num_heads (got `hidden_size`: {self.hidden_size}"
                             f" and `num_heads`: {self.total_heads}).")
        bias = getattr(config, "attention_bias", False)
        self.q_proj = nn.Linear(self.hidden_size, self.num_heads * self.head_dim, bias=bias)
        self.k_proj = nn.Linear(self.hidden_size, self.num_key_value_heads * self.head_dim, bias=bias)
        self.v_proj = nn.Linear(self.hidden_size, self.num_key_value_heads * self.head_dim, bias=bias)
        self.o_proj = nn.Linear(self.total_heads * self.head_dim, self.hidden_size, bias=bias)
        self.rotary_emb = LlamaRotaryEmbedding(self.head_dim, self.max_position_embeddings, self.rope_theta)

    # ---- cache transitions -----------------------------------------------------------------------------------
    def _prefill_cache(self, key_states, value_states):
        """Split the prompt: the first T - T % residual tokens are compressed, the tail stays fp16 (:386-434).
        V is compressed only if T > residual (:416)."""
        R = self.residual_length
        T = key_states.shape[-2]
        cc = self.compress_config
        if T % R != 0:
            if T < R:
                k_quant, k_full = None, key_states
            else:
                k_quant, k_full = key_states[:, :, :-(T % R), :].contiguous(), key_states[:, :, -(T % R):, :].contiguous()
        else:
            k_quant, k_full = key_states, None
        if k_quant is not None:
            kc, ks, km, kp, kq = key_compression(k_quant.transpose(2, 3).contiguous(), cc)
            kp, kq = [kp], [kq]
        else:
            kc = ks = km = kp = kq = None
        if T <= R:
            vc = vs = vm = vp = vq = None
            v_full = value_states
        else:
            n_quant = T - T % R
            v_full = value_states[:, :, n_quant:, :].contiguous()
            vc, vs, vm, vp, vq = value_compression(value_states[:, :, :n_quant, :].contiguous(), cc)
            vp, vq = [vp], [vq]
            if v_full.shape[-2] == 0:
                v_full = None
        return (kc, k_full, ks, km, vc, v_full, vs, vm, T, kp, kq, None, None, vp, vq, None, None)

    # ---- the fast decode path over a pre-allocated cache (round 4) ---------------------------------------------
    fast_decode = True              # class-wide switch (tests compare with the tuple path)
    decode_mask_is_zero = False     # set True when the caller's decode-step attention_mask is known to be all zeros (no padding)
    fused_rope = True               # decode steps with implicit positions: RoPE + window append in one launch (gear_rope_append)

    def _fast_eligible(self, T: int) -> bool:
        cc = self.compress_config
        R = self.residual_length
        return (self.fast_decode and self.head_dim == 128 and R in (64, 128) and cc["group_size"] in (32, 64)
                and R % cc["group_size"] == 0 and cc["quantize_bit"] in (2, 4)
                and T != R                  # (the reference strands V in fp16 for a prompt of exactly `residual` tokens, :416)
                and T < 16384)

    def _store_block(self, c, k_src, v_src, T: int):
        """Compress T tokens (K [B,H,T,128], V alike) with the hook's own operators -- the reference's order of operations and of
        random draws (key first, :265-286, then value, :335-378) -- and put the payload behind token c.n_comp of the cache."""
        cc = self.compress_config
        t0, seg = c.n_comp, c._segment_of(c.n_comp)
        kc, ks, km, kp, kq = key_compression(k_src.transpose(2, 3).contiguous(), cc)
        vc, vs, vm, vp, vq = value_compression(v_src.contiguous(), cc)
        dst = [c.kcode[..., t0 // c.fpi:(t0 + T) // c.fpi], c.kscale[..., t0 // c.group:(t0 + T) // c.group],
               c.kmn[..., t0 // c.group:(t0 + T) // c.group], c.vcode[:, :, t0:t0 + T], c.vscale[:, :, t0:t0 + T], c.vmn[:, :, t0:t0 + T]]
        src = [kc, ks, km, vc, vs, vm]
        if kp is not None:
            dst += [c.kQtok[:, :, t0:t0 + T], c.kPseg[seg], c.vQtok[:, :, t0:t0 + T], c.vPseg[seg]]
            src += [kp, kq, vq, vp]
        torch._foreach_copy_(dst, src)          # (the ten slice assignments of a boundary as one multi-tensor copy)
        c.n_comp += T

    def _fast_prefill_cache(self, key_states, value_states):
        from .cache import GearKVCache
        cc = self.compress_config
        B, T = key_states.shape[0], key_states.shape[-2]
        R = self.residual_length
        # capacity: the prompt + the tokens the caller expects to generate (config.gear_max_new_tokens, default 1024), NOT the
        # model's whole context window -- the buffers scale with batch x capacity (zero-filled: tile counts, factor segments), and
        # a short prompt at a large batch would otherwise pay for 16k tokens per sequence and layer.  A cache that fills up hands
        # over to the tuple path (materialize(), forward below), as before.
        room = max(R, int(getattr(self.config, "gear_max_new_tokens", 1024)))
        cap = min(T + room, max(int(getattr(self.config, "max_position_embeddings", 4096)), T + R))
        cap = -(-cap // 128) * 128
        # (the hook stores no outliers -- slots 11, 12, 15, 16 are None in the reference's fused path -- so a sparsity in the config,
        # the simulated path's `left`, does not reach the cache)
        c = GearKVCache(B, self.num_key_value_heads, min(cap, 16384), dict(cc, left=0.0, sparsity=0.0), key_states.device,
                        self.head_dim)
        nq = T - T % R
        if nq:
            c.seg0 = nq
            self._store_block(c, key_states[:, :, :nq], value_states[:, :, :nq], nq)
        c.n_win = T - nq
        if c.n_win:
            c.kwin[:, :, :c.n_win] = key_states[:, :, nq:]
            c.vwin[:, :, :c.n_win] = value_states[:, :, nq:]
        return GearHookCache(c, _uses_lowrank(cc), T)

    def fused_weights(self):
        """q/k/v projection weights as ONE [Nq + 2 Nkv, hidden] matrix for the fused token-step GEMV.  No second copy: the three
        nn.Linear parameters are re-pointed at row slices of the fused buffer (contiguous row ranges, so every other user of the
        modules sees ordinary weights).  Rebuilt if somebody replaced a weight's storage since."""
        q, k, v = self.q_proj.weight, self.k_proj.weight, self.v_proj.weight
        w = getattr(self, "_wqkv", None)
        nq, nk = q.shape[0], k.shape[0]
        if (w is None or q.data_ptr() != w.data_ptr() or k.data_ptr() != w[nq:].data_ptr()
                or v.data_ptr() != w[nq + nk:].data_ptr()):
            with torch.no_grad():
                w = torch.cat([q.data, k.data, v.data], 0).contiguous()
                q.data, k.data, v.data = w[:nq], w[nq:nq + nk], w[nq + nk:]
            self._wqkv = w
        return w

    def folded_qkv(self, norm_weight):
        """The fused q/k/v weight with the input RMSNorm's weight folded into its columns (fp16(W * n): the GEMV then streams the
        weights without the norm-weight loads and multiplies in its loop -- 17.4 us against 21.0 per layer on Llama-2-7B shapes).
        A second copy of the projection weights (LlamaDecoderLayer_GEAR.fold_norm_weights = False keeps to the modules' own); rebuilt
        when either parameter changed."""
        w = self.fused_weights()
        key = (w.data_ptr(), w._version, norm_weight.data_ptr(), norm_weight._version)
        if getattr(self, "_wqkv_folded_key", None) != key:
            with torch.no_grad():
                self._wqkv_folded = (w * norm_weight[None, :]).contiguous()
            self._wqkv_folded_key = key
        return self._wqkv_folded

    def decode_token_fused(self, res, delta, norm_weight, eps: float, hc: GearHookCache, dyn=None, fold: bool = False):
        """The hook's decode step for the decoder layer's fused path (batch <= 4, implicit positions, stock rotary, no bias, one
        rank): [residual add + RMSNorm + q/k/v GEMV + RoPE + window append] = ONE launch (gear_gemv_qkv_rope), fused attention over
        the packed cache, compress of a full window in place.  Returns (res + delta, attention output [B, Hq * 128], new cache) or
        None when the cache is full (the caller takes the tuple path)."""
        c = hc.cache
        B, K = res.shape
        wqkv = self.fused_weights()
        if fold:
            wqkv, norm_weight = self.folded_qkv(norm_weight), None
        if dyn is not None:
            # the launches of a captured token step (_HookGraph): position, window slot and lengths come from the device state
            # {pos, slot, T, W}; no host-side counter moves here
            res1 = torch.empty_like(res) if delta is not None else res
            q = torch.empty((B, self.num_heads, 1, self.head_dim), dtype=res.dtype, device=res.device)
            rc = L.load().gear_gemv_qkv_rope(L.ptr(res), L.ptr(delta), L.ptr(norm_weight), eps, L.ptr(wqkv), B, K,
                                             self.num_heads, self.num_key_value_heads, self.head_dim, 0, 0, c.R,
                                             float(self.rope_theta), L.ptr(dyn), L.ptr(res1) if delta is not None else None, L.ptr(q),
                                             L.ptr(c.kwin), L.ptr(c.vwin), L.stream_ptr(res))
            L.check(rc, "gear_gemv_qkv_rope")
            return res1, c.attend_dyn(q).view(B, self.num_heads * self.head_dim), None
        hc.check_live()
        if c.n_comp + c.n_win + 1 > c.Tmax:
            _warn_cache_full(c)
            return None
        kv_seq_len = hc.seq_len + 1
        res1 = torch.empty_like(res) if delta is not None else res
        q = torch.empty((B, self.num_heads, 1, self.head_dim), dtype=res.dtype, device=res.device)
        rc = L.load().gear_gemv_qkv_rope(L.ptr(res), L.ptr(delta), L.ptr(norm_weight), eps, L.ptr(wqkv), B, K,
                                         self.num_heads, self.num_key_value_heads, self.head_dim, kv_seq_len - 1, c.n_win, c.R,
                                         float(self.rope_theta), None, L.ptr(res1) if delta is not None else None, L.ptr(q),
                                         L.ptr(c.kwin), L.ptr(c.vwin), L.stream_ptr(res))
        L.check(rc, "gear_gemv_qkv_rope")
        c.n_win += 1
        out = c.attend(q)
        if c.n_win == c.R:
            self._store_block(c, c.kwin, c.vwin, c.R)
            c.n_win = 0
        return res1, out.view(B, self.num_heads * self.head_dim), GearHookCache(c, hc.lowrank, kv_seq_len)

    def _fast_decode_step(self, query_states, key_states, value_states, hc: GearHookCache, kv_seq_len: int, qkv_flat=None):
        """One token over the pre-allocated cache: window append, fused attention (packed K / V + factors + fp16 window: one call of
        gear_attn_decode_cache where the tuple path runs two GEMV pairs, the eager softmax and ~40 small ops), compress of a full window."""
        c = hc.cache
        hc.check_live()
        if c.n_comp + c.n_win + 1 > c.Tmax:
            _warn_cache_full(c)
            return None                                  # capacity: the caller continues on the tuple path
        if qkv_flat is not None:
            # RoPE at position kv_seq_len - 1 on q and k + window append in ONE launch (gear_rope_append: the reference's
            # rotary arithmetic op by op, :203-205) instead of ~25 eager ops
            query_states = c.append_rope(qkv_flat, self.num_heads, kv_seq_len - 1, float(self.rope_theta))
        else:
            c.append(key_states, value_states)
        out = c.attend(query_states)
        if c.n_win == c.R:
            self._store_block(c, c.kwin, c.vwin, c.R)
            c.n_win = 0
        return out, GearHookCache(c, hc.lowrank, kv_seq_len)

    def forward(self, hidden_states: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                position_ids: Optional[torch.LongTensor] = None, past_key_value: Optional[Tuple] = None,
                output_attentions: bool = False, use_cache: bool = False, **kwargs):
        bsz, q_len, _ = hidden_states.size()
        cc = self.compress_config
        query_states = self.q_proj(hidden_states).view(bsz, q_len, self.num_heads, self.head_dim).transpose(1, 2)
        key_states = self.k_proj(hidden_states).view(bsz, q_len, self.num_key_value_heads, self.head_dim).transpose(1, 2)
        value_states = self.v_proj(hidden_states).view(bsz, q_len, self.num_key_value_heads, self.head_dim).transpose(1, 2)

        kv_seq_len = key_states.shape[-2]
        if past_key_value is not None:
            kv_seq_len += past_key_value[8]
        n_rep = self.num_key_value_groups
        inv_norm = math.sqrt(self.head_dim)   # the reference divides (:261, :387)
        fast_ok = (isinstance(past_key_value, GearHookCache) and q_len == 1
                   and (attention_mask is None or self.decode_mask_is_zero))
        # implicit positions (the model passes none at decode: the token sits at kv_seq_len - 1) and the stock rotary module:
        # RoPE is fused with the window append below; anything else takes the reference's rotary code
        fused_rope = (fast_ok and self.fused_rope and position_ids is None and type(self.rotary_emb) is LlamaRotaryEmbedding
                      and hidden_states.dtype == torch.float16)
        if not fused_rope:
            if position_ids is None:
                position_ids = torch.arange(kv_seq_len - q_len, kv_seq_len, device=hidden_states.device).unsqueeze(0)
            cos, sin = self.rotary_emb(value_states, position_ids)
            query_states, key_states = apply_rotary_pos_emb(query_states, key_states, cos, sin)

        fast = None
        if isinstance(past_key_value, GearHookCache):
            if fast_ok:
                qkv_flat = None
                if fused_rope:
                    qkv_flat = torch.cat([query_states.reshape(bsz, -1), key_states.reshape(bsz, -1),
                                          value_states.reshape(bsz, -1)], dim=-1)
                fast = self._fast_decode_step(query_states, key_states, value_states, past_key_value, kv_seq_len, qkv_flat)
                if fast is None and fused_rope:          # (capacity fallback: rotate for the tuple path after all)
                    position_ids = torch.arange(kv_seq_len - q_len, kv_seq_len, device=hidden_states.device).unsqueeze(0)
                    cos, sin = self.rotary_emb(value_states, position_ids)
                    query_states, key_states = apply_rotary_pos_emb(query_states, key_states, cos, sin)
            if fast is None:
                past_key_value = past_key_value.materialize()   # capacity / mask / q_len: the tuple path takes over
        if fast is not None:
            attn_output, new_cache = fast
        elif past_key_value is not None:
            if q_len != 1:
                raise ValueError("decode steps take one token at a time (the packed-cache GEMV is q_len == 1)")
            (kc, k_full, ks, km, vc, v_full, vs, vm, _, kp, kq, _, _, vp, vq, _, _) = past_key_value
            att_qkquant = None
            if kc is not None:
                att_qkquant = matmul_withlrap(cc["group_size"], query_states, kc, ks, km, cc["quantize_bit"], kp, kq,
                                              type="key")
            k_full = _append(k_full, key_states, 2)
            att_qkfull = torch.matmul(query_states, _rep(k_full, n_rep).transpose(2, 3))
            attn_weights = att_qkfull if att_qkquant is None else torch.cat([att_qkquant, att_qkfull], dim=-1)
            attn_weights = attn_weights / inv_norm
            if k_full.shape[-2] == self.residual_length:            # :265 -- compress the full window
                assert self.residual_length % self.group_size == 0
                kc_n, ks_n, km_n, kp_n, kq_n = key_compression(k_full.transpose(2, 3).contiguous(), cc)
                k_full = None
                if kc is not None:
                    kc, ks, km = torch.cat([kc, kc_n], 3), torch.cat([ks, ks_n], 3), torch.cat([km, km_n], 3)
                    if kp_n is not None:
                        kp, kq = list(kp), list(kq)
                        _append_factor(kp, kp_n)
                        _append_factor(kq, kq_n)
                else:
                    kc, ks, km, kp, kq = kc_n, ks_n, km_n, [kp_n], [kq_n]
            if attn_weights.size() != (bsz, self.num_heads, q_len, kv_seq_len):
                raise ValueError(f"Attention weights should be of size {(bsz, self.num_heads, q_len, kv_seq_len)}, but is"
                                 f" {attn_weights.size()}")
            if attention_mask is not None:
                if attention_mask.size() != (bsz, 1, q_len, kv_seq_len):
                    raise ValueError(f"Attention mask should be of size {(bsz, 1, q_len, kv_seq_len)}, but is "
                                     f"{attention_mask.size()}")
                attn_weights = torch.max(attn_weights + attention_mask,
                                         torch.tensor(torch.finfo(attn_weights.dtype).min, device=attn_weights.device))
            attn_weights = F.softmax(attn_weights, dim=-1, dtype=torch.float32).to(query_states.dtype)
            v_full = _append(v_full, value_states, 2)
            value_full_length = v_full.shape[-2]
            if vc is None:
                attn_output = torch.matmul(attn_weights, _rep(v_full, n_rep))
            else:
                attn_output = matmul_withlrap(cc["group_size"], attn_weights[:, :, :, :-value_full_length].contiguous(), vc,
                                              vs, vm, cc["quantize_bit"], vp, vq, type="value")
                attn_output = attn_output + torch.matmul(attn_weights[:, :, :, -value_full_length:], _rep(v_full, n_rep))
            if value_full_length == self.residual_length:           # :335
                vc_n, vs_n, vm_n, vp_n, vq_n = value_compression(v_full.contiguous(), cc)
                v_full = None
                if vc is not None:
                    vc, vs, vm = torch.cat([vc, vc_n], 2), torch.cat([vs, vs_n], 2), torch.cat([vm, vm_n], 2)
                    if vp_n is not None:
                        vp, vq = list(vp), list(vq)
                        _append_factor(vp, vp_n)
                        _append_factor(vq, vq_n)
                else:
                    vc, vs, vm, vp, vq = vc_n, vs_n, vm_n, [vp_n], [vq_n]
            new_cache = (kc, k_full, ks, km, vc, v_full, vs, vm, kv_seq_len, kp, kq, None, None, vp, vq, None, None)
        else:
            attn_weights = torch.matmul(query_states, _rep(key_states, n_rep).transpose(2, 3)) / inv_norm
            if attn_weights.size() != (bsz, self.num_heads, q_len, kv_seq_len):
                raise ValueError(f"Attention weights should be of size {(bsz, self.num_heads, q_len, kv_seq_len)}, but is"
                                 f" {attn_weights.size()}")
            if attention_mask is not None:
                if attention_mask.size() != (bsz, 1, q_len, kv_seq_len):
                    raise ValueError(f"Attention mask should be of size {(bsz, 1, q_len, kv_seq_len)}, but is "
                                     f"{attention_mask.size()}")
                attn_weights = torch.max(attn_weights + attention_mask,
                                         torch.tensor(torch.finfo(attn_weights.dtype).min, device=attn_weights.device))
            attn_weights = F.softmax(attn_weights, dim=-1, dtype=torch.float32).to(query_states.dtype)
            attn_output = torch.matmul(attn_weights, _rep(value_states, n_rep))
            if use_cache and self._fast_eligible(kv_seq_len):
                new_cache = self._fast_prefill_cache(key_states, value_states)
            else:
                new_cache = self._prefill_cache(key_states, value_states) if use_cache else None

        if attn_output.size() != (bsz, self.num_heads, q_len, self.head_dim):
            raise ValueError(f"`attn_output` should be of size {(bsz, self.num_heads, q_len, self.head_dim)}, but is"
                             f" {attn_output.size()}")
        attn_output = attn_output.transpose(1, 2).contiguous().reshape(bsz, q_len, self.num_heads * self.head_dim)
        if self.tp_world > 1:
            from .parallel import all_gather_heads
            attn_output = all_gather_heads(attn_output, self.tp_world, self.tp_group)
        attn_output = self.o_proj(attn_output)
        return attn_output, None, (new_cache if use_cache else None)


# ------------------------------------------------------------------------------------------------- minimal Llama
class LlamaRMSNorm(nn.Module):
    def __init__(self, hidden_size, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward(self, x):
        dt = x.dtype
        x = x.to(torch.float32)
        x = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + self.variance_epsilon)
        return self.weight * x.to(dt)


class LlamaMLP(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.gate_proj = nn.Linear(config.hidden_size, config.intermediate_size, bias=False)
        self.up_proj = nn.Linear(config.hidden_size, config.intermediate_size, bias=False)
        self.down_proj = nn.Linear(config.intermediate_size, config.hidden_size, bias=False)

    def forward(self, x):
        return self.down_proj(F.silu(self.gate_proj(x)) * self.up_proj(x))


class LlamaDecoderLayer_GEAR(nn.Module):
    def __init__(self, config, layer_idx, compress_config):
        super().__init__()
        self.self_attn = LlamaAttention_GEAR(layer_idx, config, compress_config)
        self.mlp = LlamaMLP(config)
        self.input_layernorm = LlamaRMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.post_attention_layernorm = LlamaRMSNorm(config.hidden_size, eps=config.rms_norm_eps)

    def forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_value=None, use_cache=False):
        residual = hidden_states
        hidden_states, _, present = self.self_attn(self.input_layernorm(hidden_states), attention_mask, position_ids,
                                                   past_key_value, use_cache=use_cache)
        hidden_states = residual + hidden_states
        hidden_states = hidden_states + self.mlp(self.post_attention_layernorm(hidden_states))
        return hidden_states, present

    def fused_gate_up(self):
        """gate_proj / up_proj weights as ONE [2 I, hidden] matrix ([gate rows | up rows]); like the attention's q/k/v: the two
        parameters are re-pointed at slices of it, no second copy."""
        g, u = self.mlp.gate_proj.weight, self.mlp.up_proj.weight
        w = getattr(self, "_wgu", None)
        I = g.shape[0]
        if w is None or g.data_ptr() != w.data_ptr() or u.data_ptr() != w[I:].data_ptr():
            with torch.no_grad():
                w = torch.cat([g.data, u.data], 0).contiguous()
                g.data, u.data = w[:I], w[I:]
            self._wgu = w
        return w

    fold_norm_weights = True        # decode steps: norm weights folded into second copies of q/k/v and gate/up (+ ~9 GB at 7B)

    def folded_gate_up(self):
        """fused_gate_up() with post_attention_layernorm's weight folded into its columns (a second copy; see folded_qkv)."""
        w, n = self.fused_gate_up(), self.post_attention_layernorm.weight
        key = (w.data_ptr(), w._version, n.data_ptr(), n._version)
        if getattr(self, "_wgu_folded_key", None) != key:
            with torch.no_grad():
                self._wgu_folded = (w * n[None, :]).contiguous()
            self._wgu_folded_key = key
        return self._wgu_folded

    def fused_step_ok(self, B: int, past) -> bool:
        """The layer's token step can run as the six fused launches (batch <= 4, pre-allocated cache, one rank, stock rotary, no biases)."""
        at, mlp = self.self_attn, self.mlp
        return (B <= 4 and isinstance(past, GearHookCache) and at.tp_world == 1 and at.q_proj.bias is None and at.fused_rope
                and at.fast_decode and type(at.rotary_emb) is LlamaRotaryEmbedding and mlp.down_proj.bias is None
                and mlp.gate_proj.weight.shape[0] % 2 == 0)

    def decode_step(self, res: torch.Tensor, delta: Optional[torch.Tensor], past, dyn=None):
        """One token through the layer with the glue fused (csrc/decode_ops.hip; the same fp16 arithmetic op by op as the modules
        above): res [B, hidden] is the residual stream BEFORE `delta` (the previous layer's MLP output, or None) is added.
        [residual add + RMSNorm] -> attention hook (its reference signature) -> [residual add + RMSNorm] -> gate / up ->
        [SiLU * up] -> down: 15 launches where forward() issues ~50.  Returns (res', delta', present)."""
        lib = L.load()
        B, Hd = res.shape
        st = L.stream_ptr(res)
        ln1, ln2 = self.input_layernorm, self.post_attention_layernorm
        at, mlp = self.self_attn, self.mlp
        if self.fused_step_ok(B, past):
            # 6 launches: [add + norm + qkv + RoPE + append] -> attention (2) -> [o_proj + add] -> [norm + gate/up + SiLU * up]
            # -> [down + add]; the weights are the modules' own (q/k/v and gate/up fused by re-pointing the parameters at slices of
            # one buffer: gate rows, then up rows -- the GEMV pairs row j with row I + j)
            fold = self.fold_norm_weights
            r = at.decode_token_fused(res, delta, ln1.weight, ln1.variance_epsilon, past, dyn, fold)
            if r is not None:
                res1, a, present = r
                res2 = torch.empty_like(res)
                L.check(lib.gear_gemv_f16_add(L.ptr(a), L.ptr(at.o_proj.weight), B, a.shape[1], Hd, L.ptr(res1), L.ptr(res2), st),
                        "gear_gemv_f16_add")
                wgu = self.folded_gate_up() if fold else self.fused_gate_up()
                I = wgu.shape[0] // 2
                act = torch.empty((B, I), dtype=res.dtype, device=res.device)
                L.check(lib.gear_gemv_f16_norm(L.ptr(res2), None, None if fold else L.ptr(ln2.weight), ln2.variance_epsilon, L.ptr(wgu),
                                               B, Hd, 2 * I, 2, None, L.ptr(act), st), "gear_gemv_f16_norm")
                res3 = torch.empty_like(res)
                L.check(lib.gear_gemv_f16_add(L.ptr(act), L.ptr(mlp.down_proj.weight), B, I, Hd, L.ptr(res2), L.ptr(res3), st),
                        "gear_gemv_f16_add")
                return res3, None, present
        x, res1 = torch.empty_like(res), torch.empty_like(res)
        L.check(lib.gear_add_rmsnorm(L.ptr(res), L.ptr(delta), L.ptr(ln1.weight), B, Hd, ln1.variance_epsilon, L.ptr(res1),
                                     L.ptr(x), st), "gear_add_rmsnorm")
        a, _, present = self.self_attn(x.unsqueeze(1), None, None, past, use_cache=True)
        x2, res2 = torch.empty_like(res), torch.empty_like(res)
        L.check(lib.gear_add_rmsnorm(L.ptr(res1), L.ptr(a.reshape(B, Hd)), L.ptr(ln2.weight), B, Hd, ln2.variance_epsilon,
                                     L.ptr(res2), L.ptr(x2), st), "gear_add_rmsnorm")
        I = mlp.gate_proj.weight.shape[0]
        if B == 1:
            gu = torch.empty((1, 2 * I), dtype=res.dtype, device=res.device)
            torch.mm(x2, mlp.gate_proj.weight.t(), out=gu[:, :I])
            torch.mm(x2, mlp.up_proj.weight.t(), out=gu[:, I:])
        else:
            gu = torch.cat([mlp.gate_proj(x2), mlp.up_proj(x2)], dim=-1)
        act = torch.empty((B, I), dtype=res.dtype, device=res.device)
        L.check(lib.gear_silu_mul(L.ptr(gu), B, I, L.ptr(act), st), "gear_silu_mul")
        return res2, mlp.down_proj(act), present


class _HookGraph:
    """The decode step of LlamaModel_GEAR over GearHookCache layers as ONE hipGraph: embedding -> per layer the six launches of
    LlamaDecoderLayer_GEAR.decode_step with position / window slot / lengths read from a device-side {pos, slot, T, W} state that
    every layer's cache shares -> final RMSNorm -> state advance.  Same kernels, same order, same arithmetic as the eager fused
    step (the reference-trace tests run through it); what disappears is ~200 ctypes calls and ~200 tensor allocations of host work
    per token, which is what the eager hook step was bound by (282 tokens/s against FastGearDecoder's 340 in round 5: VERDICT r5
    item 6).  Block boundaries (every `residual` tokens) run eagerly between replays, exactly as the eager step does them.
    A graph belongs to one set of caches (one prompt): LlamaModel_GEAR builds it at the third decode step of a generation."""

    def __init__(self, model, caches, B: int):
        self.caches = caches
        dev = model.embed_tokens.weight.device
        self.state = torch.zeros(4, dtype=torch.int32, device=dev)
        self.tok = torch.zeros((B,), dtype=torch.long, device=dev)
        for c in caches:
            c.state = self.state
        self.key = tuple(id(c) for c in caches)
        self._sync()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            res, delta = model.embed_tokens(self.tok), None
            for layer, c in zip(model.layers, caches):
                res, delta, _ = layer.decode_step(res, delta, _DynPast(c), self.state)
            y = torch.empty_like(res)
            L.check(L.load().gear_add_rmsnorm(L.ptr(res), L.ptr(delta), L.ptr(model.norm.weight), B, res.shape[1],
                                              model.norm.variance_epsilon, None, L.ptr(y), L.stream_ptr(res)), "gear_add_rmsnorm")
            L.check(L.load().gear_decode_state_advance(L.ptr(self.state), L.stream_ptr(res)), "gear_decode_state_advance")
        self.graph, self.y = g, y

    def _sync(self):
        c = self.caches[0]
        self.state.copy_(torch.tensor([c.n_comp + c.n_win, c.n_win, c.n_comp, c.n_win + 1], dtype=torch.int32))

    def step(self, model, token, pasts):
        """token [B] -> final hidden state [B, hidden] (a static buffer) and the layers' new GearHookCache objects."""
        self.tok.copy_(token)
        self.graph.replay()
        presents = []
        boundary = False
        for layer, c, hc in zip(model.layers, self.caches, pasts):
            c.n_win += 1
            if c.n_win == c.R:
                layer.self_attn._store_block(c, c.kwin, c.vwin, c.R)
                c.n_win = 0
                boundary = True
            presents.append(GearHookCache(c, hc.lowrank, hc.seq_len + 1))
        if boundary:
            self._sync()
        return self.y, tuple(presents)


class _DynPast(GearHookCache):
    """What a captured step hands to LlamaDecoderLayer_GEAR.decode_step in place of the caller's GearHookCache: only the cache."""

    def __init__(self, cache):
        self.cache = cache


class LlamaModel_GEAR(nn.Module):
    fused_decode_glue = True        # decode steps over GearHookCache layers use LlamaDecoderLayer_GEAR.decode_step
    graph_decode = False            # ... replayed as one hipGraph from the third decode step of a generation on (_HookGraph).
                                    # Off: measured SLOWER than the eager launches (304 against 313 tokens/s, 7B at 4k) -- the
                                    # step is bound by its kernels, not by the host, and the captured attention runs over the
                                    # cache's capacity instead of its length

    def __init__(self, config, compress_config):
        super().__init__()
        self.config = config
        self.embed_tokens = nn.Embedding(config.vocab_size, config.hidden_size)
        self.layers = nn.ModuleList([LlamaDecoderLayer_GEAR(config, i, compress_config)
                                     for i in range(config.num_hidden_layers)])
        self.norm = LlamaRMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self._hook_graph = None
        self._hook_steps = (None, 0)

    prefetch_bases = True           # draw the next block boundary's power-iteration bases during the token steps before it

    def _prefetch_bases(self, presents):
        """The next block boundary asks, layer by layer, for one basis per K block (headwise_lrap on E^T [B,H,128,R]: P0 [B,H,R,rank])
        and one per V block (E [B,H,R,128]: P0 [B,H,128,rankv]) from torch's CPU generator (new_pack.py:296-297) -- 3 M random
        numbers for 32 layers, 6 ms of host time during which the GPU waits.  Draw them in that very order a few per token step
        instead (compress.prefetch_p0): same values, same order in the generator's stream."""
        if not self.prefetch_bases or not presents or not all(isinstance(p, GearHookCache) and p.lowrank for p in presents):
            return
        c0 = presents[0].cache
        n_layers = len(presents)
        from . import compress as Cm
        if getattr(self, "_pf_owner", None) != id(c0):
            # another generation's caches: whatever was drawn ahead for the old ones must not be handed to these
            Cm._p0_queue.clear()
            self._pf_owner, self._pf_done = id(c0), 0
        if c0.n_win == 0:
            self._pf_done = 0                      # a boundary has just consumed its bases (or the window is empty after the prefill)
        done = getattr(self, "_pf_done", 0)
        if done >= n_layers or c0.n_win == 0:
            return
        per_step = -(-n_layers // max(1, c0.R // 2))          # all drawn by the middle of the window
        for li in range(done, min(n_layers, done + per_step)):
            c = presents[li].cache
            cc = self.layers[li].self_attn.compress_config
            Cm.prefetch_p0([(c.B, c.H, c.D, c.R, int(cc["rank"])), (c.B, c.H, c.R, c.D, int(cc["rankv"]))], c.kwin.device)
        self._pf_done = min(n_layers, done + per_step)

    def _graph_step(self, input_ids, past_key_values):
        """The captured token step when it applies (see _HookGraph), else None."""
        if not self.graph_decode or torch.is_grad_enabled() or torch.cuda.is_current_stream_capturing():
            return None
        bsz = input_ids.shape[0]
        caches = [pk.cache for pk in past_key_values]
        c0 = caches[0]
        key = tuple(id(c) for c in caches)
        if any((c.n_comp, c.n_win, c.R, c.Tmax) != (c0.n_comp, c0.n_win, c0.R, c0.Tmax) for c in caches):
            return None
        if c0.n_comp + c0.n_win + 1 > c0.Tmax or not all(l.fused_step_ok(bsz, pk) for l, pk in zip(self.layers, past_key_values)):
            return None
        for pk in past_key_values:
            pk.check_live()
        g = self._hook_graph
        if g is None or g.key != key:
            # a new generation (new caches): two eager steps first (short generations never pay for a capture), then the graph
            last, n = self._hook_steps
            n = n + 1 if last == key else 1
            self._hook_steps = (key, n)
            if n < 3:
                return None
            g = self._hook_graph = _HookGraph(self, caches, bsz)
        elif (int(c0.n_comp + c0.n_win), c0.n_win) != getattr(g, "_expect", (int(c0.n_comp + c0.n_win), c0.n_win)):
            g._sync()                  # somebody stepped these caches eagerly in between
        y, presents = g.step(self, input_ids[:, 0], past_key_values)
        g._expect = (int(c0.n_comp + c0.n_win), c0.n_win)
        return y.unsqueeze(1), presents

    def forward(self, input_ids, past_key_values=None, use_cache=True):
        bsz, q_len = input_ids.shape
        if (q_len == 1 and use_cache and past_key_values is not None and self.fused_decode_glue
                and all(isinstance(pk, GearHookCache) for pk in past_key_values)
                and self.embed_tokens.weight.dtype == torch.float16 and self.embed_tokens.weight.is_cuda
                and self.config.hidden_size % 8 == 0 and self.config.hidden_size <= 8192
                and self.layers[0].mlp.gate_proj.bias is None):
            r = self._graph_step(input_ids, past_key_values)
            if r is not None:
                return r
            # decode step over pre-allocated caches: fused glue around the attention hook (LlamaDecoderLayer_GEAR.decode_step)
            res, delta = self.embed_tokens(input_ids[:, 0]), None
            presents = []
            for layer, pk in zip(self.layers, past_key_values):
                res, delta, present = layer.decode_step(res, delta, pk)
                presents.append(present)
            y = torch.empty_like(res)
            L.check(L.load().gear_add_rmsnorm(L.ptr(res), L.ptr(delta), L.ptr(self.norm.weight), bsz, res.shape[1],
                                              self.norm.variance_epsilon, None, L.ptr(y), L.stream_ptr(res)), "gear_add_rmsnorm")
            self._prefetch_bases(presents)
            return y.unsqueeze(1), tuple(presents)
        if past_key_values is None:
            from . import compress as Cm
            Cm._p0_queue.clear()                   # a new prompt: no basis drawn ahead for an earlier one survives
            self._pf_owner = None
        past_len = past_key_values[0][8] if past_key_values is not None else 0     # slot 8 (:624)
        position_ids = torch.arange(past_len, past_len + q_len, device=input_ids.device).unsqueeze(0)
        hidden_states = self.embed_tokens(input_ids)
        mask = None
        if q_len > 1:
            mask = torch.full((q_len, q_len), torch.finfo(hidden_states.dtype).min, device=input_ids.device,
                              dtype=hidden_states.dtype).triu(1)[None, None].expand(bsz, 1, q_len, q_len)
        presents = []
        for i, layer in enumerate(self.layers):
            hidden_states, present = layer(hidden_states, mask, position_ids,
                                           past_key_values[i] if past_key_values is not None else None, use_cache)
            presents.append(present)
        return self.norm(hidden_states), tuple(presents)


class LlamaForCausalLM_GEARKIVI(nn.Module):
    """modeling_llamagear.py:711 -- causal LM over LlamaModel_GEAR with a greedy generate() for the timing harness
    (cuda_supported_gear/test.py:95-102).  Weights are whatever the caller loads; the bench uses random init."""

    def __init__(self, config, compress_config=None):
        super().__init__()
        self.config = config
        self.model = LlamaModel_GEAR(config, compress_config)
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False)

    def forward(self, input_ids, past_key_values=None, use_cache=True):
        hidden, presents = self.model(input_ids, past_key_values, use_cache)
        h = hidden[:, -1:, :]
        w = self.lm_head.weight
        if (h.shape[0] <= 4 and h.is_cuda and h.dtype == torch.float16 and w.dtype == torch.float16 and w.is_contiguous()
                and h.shape[-1] % 8 == 0 and not torch.is_grad_enabled()):
            # one token: the weight-streaming GEMV of the token step (csrc/gemv_f16.hip) instead of the library GEMM
            x = h.reshape(h.shape[0], -1).contiguous()
            y = torch.empty((x.shape[0], w.shape[0]), dtype=h.dtype, device=h.device)
            L.check(L.load().gear_gemv_f16(L.ptr(x), L.ptr(w), x.shape[0], x.shape[1], w.shape[0], L.ptr(y), L.stream_ptr(x)), "gear_gemv_f16")
            return y.unsqueeze(1), presents
        return self.lm_head(h), presents

    @torch.no_grad()
    def generate(self, input_ids, max_length: int, use_cache: bool = True):
        logits, past = self.forward(input_ids, None, True)
        out = [input_ids]
        nxt = logits[:, -1].argmax(-1, keepdim=True)
        out.append(nxt)
        while sum(t.shape[1] for t in out) < max_length:
            logits, past = self.forward(nxt, past, True)
            nxt = logits[:, -1].argmax(-1, keepdim=True)
            out.append(nxt)
        return torch.cat(out, dim=1)
