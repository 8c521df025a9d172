"""Head-wise sharding of the GEAR cache + attention across GPUs (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on MI355X, "gloo" in the CPU tests).

The reference has no distributed code at all; this is the build's design for
BASELINE.json's 1/2/4/8-GPU rows:
  * KV heads are split contiguously across ranks; every quantization group, low-rank factor pair and K outlier row
    lives inside one head, so compress / decompress need no communication.  The one exception is the simulated
    path's V outlier selection, a top-k over the whole token row ACROSS heads (compress_function.py:304-311): a shard
    selects k / world per side inside its own heads (documented divergence for world > 1; exact for world == 1).
  * softmax is per head, so attention is local; the only exchange is an all-gather of the per-rank attention output
    [B, q, H_local*D] (a few KiB per layer per token: latency-bound single hop over xGMI), after which every rank
    applies the replicated o_proj.
"""
from __future__ import annotations

from typing import Tuple

import torch


def shard_heads(n_heads: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous [begin, end) head range of `rank`."""
    if n_heads % world:
        raise ValueError(f"{n_heads} heads do not divide across {world} ranks")
    per = n_heads // world
    return rank * per, (rank + 1) * per


def outliers_per_shard(k_full: int, world: int) -> int:
    """Per-side outlier count of a V token row restricted to one shard's heads (k scales with the row length)."""
    return max(1, k_full // world) if k_full > 0 else 0


def all_gather_heads(x: torch.Tensor, world: int, group=None) -> torch.Tensor:
    """x [B, q, H_local*D] on every rank -> [B, q, world*H_local*D] with rank r's heads at slot r."""
    import torch.distributed as dist
    x = x.contiguous()
    out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)   # ranks concatenated on dim 0
    with torch.no_grad():                                 # (inference path; gloo's implementation writes through views)
        dist.all_gather_into_tensor(out, x.detach(), group=group)   # one collective into one buffer (no tensor list, no cat)
    return out.view((world,) + tuple(x.shape)).movedim(0, -2).reshape(*x.shape[:-1], world * x.shape[-1])


class HeadGather:
    """All-gather of the per-rank attention output of a decode step into a PRE-ALLOCATED buffer with
    all_gather_into_tensor: one collective per layer per token, no list of tensors, no torch.cat (the exchange is a few KiB
    and latency-bound: a single hop over xGMI on MI355X, where backend "nccl" is RCCL).
    x [B, H_local*D] on every rank -> [B, world*H_local*D] with rank r's heads at slot r."""

    def __init__(self, world: int, batch: int, width: int, dtype, device, group=None):
        import torch.distributed as dist
        self.world, self.B, self.width, self.group = world, batch, width, group
        self.buf = torch.empty((world * batch, width), dtype=dtype, device=device)       # ranks concatenated on dim 0
        # gloo cannot gather device tensors: the one-GPU debug mode of bench.py / the tests stage through the host
        self.stage = dist.get_backend(group) == "gloo" and torch.device(device).type == "cuda"

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        import torch.distributed as dist
        if self.stage:
            hb = torch.empty(self.buf.shape, dtype=self.buf.dtype)
            dist.all_gather_into_tensor(hb, x.contiguous().cpu(), group=self.group)
            self.buf.copy_(hb)
        else:
            dist.all_gather_into_tensor(self.buf, x.contiguous(), group=self.group)
        if self.B == 1:
            return self.buf.view(1, self.world * self.width)          # rank-major == head-major for one row
        return self.buf.view(self.world, self.B, self.width).permute(1, 0, 2).reshape(self.B, self.world * self.width)


def shard_attention_weights(full_attn, local_attn):
    """Copy the rank-local slices of a full LlamaAttention_GEAR's projections into a head-sharded instance
    (q/k/v are column-parallel over heads, o_proj is replicated)."""
    D = full_attn.head_dim
    r, w = local_attn.tp_rank, local_attn.tp_world
    qb, qe = shard_heads(full_attn.total_heads, w, r)
    kb, ke = shard_heads(full_attn.config.num_key_value_heads, w, r)
    with torch.no_grad():
        local_attn.q_proj.weight.copy_(full_attn.q_proj.weight[qb * D:qe * D])
        local_attn.k_proj.weight.copy_(full_attn.k_proj.weight[kb * D:ke * D])
        local_attn.v_proj.weight.copy_(full_attn.v_proj.weight[kb * D:ke * D])
        local_attn.o_proj.weight.copy_(full_attn.o_proj.weight)
        if full_attn.q_proj.bias is not None:
            local_attn.q_proj.bias.copy_(full_attn.q_proj.bias[qb * D:qe * D])
            local_attn.k_proj.bias.copy_(full_attn.k_proj.bias[kb * D:ke * D])
            local_attn.v_proj.bias.copy_(full_attn.v_proj.bias[kb * D:ke * D])
            local_attn.o_proj.bias.copy_(full_attn.o_proj.bias)
    return local_attn
