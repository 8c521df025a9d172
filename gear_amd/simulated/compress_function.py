"""Drop-in counterpart of GenerationBench/GenerationTest/GEARLM/Simulated/compress_function.py.

The reference *simulates* compression (quantize -> dequantize in place on the fp16 cache, torch eager).  Here the
same function names run the REAL thing on the GPU: compress to the packed payload with the HIP kernels
(gear_amd.compress) and decompress it again, returning the lossy fp16 tensor the caller expects.  Parity target
(BASELINE.json north_star): within 1e-3 relative of the reference's simulated output on identical K/V; bit-exact
where the result is a pure quantize/dequantize (KIVI_V2).

Not reproduced: token_preserving windows (start_saving / locality_saving, :433-438) and group sizes that span heads
(the KCVT variants, :441-452, :496-525, :555-582) -- both raise NotImplementedError.
"""
from __future__ import annotations

import torch

from .. import compress as C


def _need_head_local(group_size, D, what):
    if group_size > D or D % group_size:
        raise NotImplementedError(f"{what}: group_size {group_size} must divide head_dim {D} "
                                  "(groups spanning heads -- the KCVT variants -- are not built)")


def _half(x):
    if x.dtype == torch.float16:
        return x
    h = x.half()
    if not torch.equal(h.to(x.dtype), x):
        raise NotImplementedError("inputs must be fp16-representable (the KV cache is fp16)")
    return h


# ------------------------------------------------------------------------------------------------ a9
def fake_groupwise_token_asymmetric_quantization(input: torch.Tensor, quantize_bit, group_size=128):
    """compress_function.py:7-37: per-token groups along H*D, fp32 arithmetic, result in input.dtype."""
    B, H, T, D = input.shape
    if (H * D) % group_size:
        raise ValueError("group_size should be a factor of the last dimension size")   # :16-17
    _need_head_local(group_size, D, "token quantization")
    p = C.compress_value(_half(input), quantize_bit, group_size, mode="fp32")
    return C.decompress(p).type(input.dtype)


def fake_groupwise_channel_asymmetric_quantization_new(input: torch.Tensor, quantize_bit, group_size=128):
    """compress_function.py:39-67: per-channel groups of `group_size` tokens, arithmetic in the INPUT dtype."""
    B, H, T, D = input.shape
    assert T % group_size == 0
    mode = "fp16" if input.dtype == torch.float16 else "fp32"
    p = C.compress_key(_half(input), quantize_bit, group_size, mode=mode)
    return C.decompress(p).type(input.dtype)


def fake_poweriteration_group(input: torch.Tensor, loop, rank, device, p_base, q_base):
    """compress_function.py:69-98: returns the rank-`rank` reconstruction Q P^T in input.dtype.
    p_base: optional list [tensor float [B,H,D,rank]] (the reference's list idiom) or None (drawn like the reference)."""
    B, H, T, D = input.shape
    P0 = p_base[0] if p_base is not None else C.draw_p0(B, H, T, D, rank, input.device)
    E = input if input.dtype in (torch.float16, torch.float32) else input.float()
    P, Q = C.lowrank(E, rank, loop, P0, transposed=False, out_dtype=torch.float32)
    return torch.matmul(Q, P.transpose(2, 3)).type(input.dtype)


# ------------------------------------------------------------------------------------------------ a11
def gears_channelQ(input, quantize_bit, group_size=128, sparsity=0.0):
    """compress_function.py:261-296: K outliers per channel row + fp32 channel quantization -> fp16."""
    B, H, T, D = input.shape
    k = C.outlier_count(B, H, T, D, sparsity)
    p = C.compress_key(_half(input), quantize_bit, group_size, k_out=k, mode="fp32")
    return C.decompress(p)


def gears_tokenQ(input, quantize_bit, group_size=128, sparsity=0.0):
    """compress_function.py:297-333: V outliers per token row (across heads) + fp32 token quantization -> fp16."""
    B, H, T, D = input.shape
    _need_head_local(group_size, D, "token quantization")
    k = C.outlier_count(B, H, T, D, sparsity)
    p = C.compress_value(_half(input), quantize_bit, group_size, k_out=k, mode="fp32")
    return C.decompress(p)


# ------------------------------------------------------------------------------------------------ a12
def gearslkivi_channelQ_new(input, quantize_bit, group_size=128, sparsity=0.0, rank=0, loop=1, P0=None):
    """compress_function.py:213-220 (GEAR, K): outliers + quant + low-rank of the residual."""
    B, H, T, D = input.shape
    k = C.outlier_count(B, H, T, D, sparsity)
    p = C.compress_key(_half(input), quantize_bit, group_size, k_out=k, rank=rank, loop=loop, mode="fp32", P0=P0)
    return C.decompress(p)


def gearslkivi_tokenQ_new(input, quantize_bit, group_size=128, sparsity=0.0, rank=0, loop=1, P0=None):
    """compress_function.py:204-211 (GEAR, V)."""
    B, H, T, D = input.shape
    _need_head_local(group_size, D, "token quantization")
    k = C.outlier_count(B, H, T, D, sparsity)
    p = C.compress_value(_half(input), quantize_bit, group_size, k_out=k, rank=rank, loop=loop, mode="fp32", P0=P0)
    return C.decompress(p)


def tokenwise_gearlkivi_channelQ(input, quantize_bit, group_size=128, r=0, loop=1, P0=None):
    """compress_function.py:334-357 (GEARL, K): input-dtype channel quantization + low-rank."""
    mode = "fp16" if input.dtype == torch.float16 else "fp32"
    p = C.compress_key(_half(input), quantize_bit, group_size, rank=r, loop=loop, mode=mode, P0=P0)
    return C.decompress(p).type(input.dtype)


def tokenwise_gearlkivi_tokenQ(input, quantize_bit, group_size=128, r=0, loop=1, P0=None):
    """compress_function.py:399-418 (GEARL, V): fp32 token quantization + low-rank."""
    B, H, T, D = input.shape
    _need_head_local(group_size, D, "token quantization")
    p = C.compress_value(_half(input), quantize_bit, group_size, rank=r, loop=loop, mode="fp32", P0=P0)
    return C.decompress(p).type(input.dtype)


def compress_insert_function(previous_key, previous_value, compress_config, layer_idx, pbase1=None, qbase1=None,
                             pbase2=None, qbase2=None, prefill=None):
    """compress_function.py:421-584.  previous_key / previous_value fp16 [B,H,T,D] -> (key, value) lossy fp16.

    compress_config: gear_amd.simulated.CompressionConfig after copy_for_all_attention() (per-layer lists), or any
    object with the same attributes.  Methods: KIVI_V2, GEAR, GEARL (the others named in the reference's dispatcher
    raise NotImplementedError; unknown names are a no-op exactly like the reference).
    pbase1 / pbase2 (unused by the reference's body): optional initial bases [B,H,D,rank] for K / V, so that callers
    can make runs reproducible; by default they are drawn like the reference (CPU generator, K first)."""
    batch, num_head, seq_len, sep_dim = previous_key.shape
    if compress_config.token_preserving[layer_idx] == True:  # noqa: E712
        raise NotImplementedError("token_preserving windows are not built")
    method = compress_config.compress_method[layer_idx]
    bits = compress_config.quantize_bit[layer_idx]
    group = compress_config.group_size[layer_idx]
    if method == "KIVI_V2":
        previous_key = fake_groupwise_channel_asymmetric_quantization_new(previous_key, bits, group)
        previous_value = fake_groupwise_token_asymmetric_quantization(previous_value, bits, group)
    elif method in ("GEAR", "GEARL"):
        if prefill is True:
            rank_used = int(compress_config.prefill_rank[layer_idx])
            rankv_used = int(compress_config.prefill_rankv[layer_idx])
        else:
            rank_used = int(compress_config.rank[layer_idx])
            rankv_used = int(compress_config.rankv[layer_idx])
        loop = compress_config.loop[layer_idx]
        if method == "GEAR":
            left = compress_config.left[layer_idx]
            previous_key = gearslkivi_channelQ_new(previous_key, bits, group, left, rank_used, loop, P0=pbase1)
            previous_value = gearslkivi_tokenQ_new(previous_value, bits, group, left, rankv_used, loop, P0=pbase2)
        else:
            previous_key = tokenwise_gearlkivi_channelQ(previous_key, bits, group, rank_used, loop, P0=pbase1)
            previous_value = tokenwise_gearlkivi_tokenQ(previous_value, bits, group, rankv_used, loop, P0=pbase2)
    elif method in ("KCVT", "GEAR-KCVT", "GEARL-KCVT"):
        raise NotImplementedError(f"{method}: groups spanning the whole sequence / all heads are not built")
    return previous_key, previous_value
