"""Counterpart of cuda_supported_gear/modeling_llama_kivi.py (the KIVI baseline the reference's timing harness compares
against, cuda_supported_gear/test.py:25-62) and of the Mistral-shaped model (GenerationBench/.../Simulated/modeling_mistral.py:
679-762 is the same attention with grouped KV heads and a sliding-window mask).

LlamaAttention_KIVI.forward (:81-289), same 9-slot cache tuple (:268):
  0 K code int32 [B,Hkv,D,Tq/fpi] (K^T, packed along tokens)   1 K_full fp16 [B,Hkv,t<R,D] | None   2 K scale   3 K mn
  4 V code int32 [B,Hkv,Tv,D/fpi]                              5 V_full fp16 [B,Hkv,<=R,D]           6 V scale   7 V mn
  8 kv_seq_len
State machine -- different from the GEAR hook on the V side:
  K: an fp16 window that is quantized per channel as a whole block when it holds `residual_length` tokens (:149-162), the prompt
     split at T - T % R (:222-236);
  V: a SLIDING fp16 window of the `residual_length` most recent tokens; once it holds R + 1, the OLDEST single token is quantized
     per token and appended to the packed part (:200-213); the prompt keeps its last R tokens in fp16 (:238-248).
No low-rank factors, no outliers.  All quantize / dequant-GEMV work is on the HIP kernels (quant_pack.hip, gemv.hip); GQA is
supported (the reference asserts num_key_value_groups == 1, :131)."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Tuple

import torch
import torch.nn.functional as F

from .modeling_llamagear import (LlamaAttention_GEAR, LlamaConfigLite, LlamaForCausalLM_GEARKIVI, _append, _rep,
                                 apply_rotary_pos_emb)
from .quant.matmul import cuda_bmm_fA_qB_outer
from .quant.new_pack import triton_quantize_and_pack_along_last_dim


class LlamaAttention_KIVI(LlamaAttention_GEAR):
    """modeling_llama_kivi.py:40-289."""

    def __init__(self, layer_idx, config, compress_config=None, **kw):
        cc = dict(compress_config or {})
        cc.setdefault("residual", config.residual_length)
        cc.setdefault("compress_method", "KIVI")
        super().__init__(layer_idx, config, cc, **kw)

    def _prefill_cache_kivi(self, key_states, value_states):
        R, T = self.residual_length, key_states.shape[-2]
        if T % R != 0:
            if T < R:
                k_quant, k_full = None, key_states
            else:
                k_quant, k_full = key_states[:, :, :-(T % R), :].contiguous(), key_states[:, :, -(T % R):, :].contiguous()
        else:
            k_quant, k_full = key_states, None
        if k_quant is not None:
            kc, ks, km = triton_quantize_and_pack_along_last_dim(k_quant.transpose(2, 3).contiguous(), self.group_size, self.k_bits)
        else:
            kc = ks = km = None
        if T <= R:
            vc = vs = vm = None
            v_full = value_states
        else:
            v_full = value_states[:, :, -R:, :].contiguous()
            vc, vs, vm = triton_quantize_and_pack_along_last_dim(value_states[:, :, :-R, :].contiguous(), self.group_size, self.v_bits)
        return (kc, k_full, ks, km, vc, v_full, vs, vm, T)

    def forward(self, hidden_states: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                position_ids: Optional[torch.LongTensor] = None, past_key_value: Optional[Tuple] = None,
                output_attentions: bool = False, use_cache: bool = False, **kwargs):
        bsz, q_len, _ = hidden_states.size()
        query_states = self.q_proj(hidden_states).view(bsz, q_len, self.num_heads, self.head_dim).transpose(1, 2)
        key_states = self.k_proj(hidden_states).view(bsz, q_len, self.num_key_value_heads, self.head_dim).transpose(1, 2)
        value_states = self.v_proj(hidden_states).view(bsz, q_len, self.num_key_value_heads, self.head_dim).transpose(1, 2)
        kv_seq_len = key_states.shape[-2]
        if past_key_value is not None:
            kv_seq_len += past_key_value[-1]
        if position_ids is None:
            position_ids = torch.arange(kv_seq_len - q_len, kv_seq_len, device=hidden_states.device).unsqueeze(0)
        cos, sin = self.rotary_emb(value_states, position_ids)
        query_states, key_states = apply_rotary_pos_emb(query_states, key_states, cos, sin)
        n_rep, R = self.num_key_value_groups, self.residual_length

        def masked_softmax(w):
            if w.size() != (bsz, self.num_heads, q_len, kv_seq_len):
                raise ValueError(f"Attention weights should be of size {(bsz, self.num_heads, q_len, kv_seq_len)}, but is {w.size()}")
            if attention_mask is not None:
                if attention_mask.size() != (bsz, 1, q_len, kv_seq_len):
                    raise ValueError(f"Attention mask should be of size {(bsz, 1, q_len, kv_seq_len)}, but is {attention_mask.size()}")
                w = torch.max(w + attention_mask, torch.tensor(torch.finfo(w.dtype).min, device=w.device))
            return F.softmax(w, dim=-1, dtype=torch.float32).to(query_states.dtype)

        if past_key_value is not None:
            if q_len != 1:
                raise ValueError("decode steps take one token at a time (the packed-cache GEMV is q_len == 1)")
            kc, k_full, ks, km, vc, v_full, vs, vm, _ = past_key_value
            att_qkquant = cuda_bmm_fA_qB_outer(self.group_size, query_states, kc, ks, km, self.k_bits) if kc is not None else None
            k_full = _append(k_full, key_states, 2)
            att_qkfull = torch.matmul(query_states, _rep(k_full, n_rep).transpose(2, 3))
            w = att_qkfull if att_qkquant is None else torch.cat([att_qkquant, att_qkfull], dim=-1)
            w = w / math.sqrt(self.head_dim)
            if k_full.shape[-2] == R:                                  # :149 -- the K window is full: quantize the block
                assert R % self.group_size == 0
                kc_n, ks_n, km_n = triton_quantize_and_pack_along_last_dim(k_full.transpose(2, 3).contiguous(), self.group_size,
                                                                           self.k_bits)
                k_full = None
                if kc is not None:
                    kc, ks, km = torch.cat([kc, kc_n], 3), torch.cat([ks, ks_n], 3), torch.cat([km, km_n], 3)
                else:
                    kc, ks, km = kc_n, ks_n, km_n
            w = masked_softmax(w)
            v_full = torch.cat([v_full, value_states], dim=2)
            nfull = v_full.shape[-2]
            if vc is None:
                attn_output = torch.matmul(w, _rep(v_full, n_rep))
            else:
                attn_output = cuda_bmm_fA_qB_outer(self.group_size, w[:, :, :, :-nfull].contiguous(), vc, vs, vm, self.v_bits)
                attn_output = attn_output + torch.matmul(w[:, :, :, -nfull:], _rep(v_full, n_rep))
            if nfull > R:                                              # :200 -- slide the V window: quantize its oldest token
                assert nfull == R + 1
                vc_n, vs_n, vm_n = triton_quantize_and_pack_along_last_dim(v_full[:, :, :1, :].contiguous(), self.group_size,
                                                                           self.v_bits)
                v_full = v_full[:, :, 1:, :].contiguous()
                if vc is not None:
                    vc, vs, vm = torch.cat([vc, vc_n], 2), torch.cat([vs, vs_n], 2), torch.cat([vm, vm_n], 2)
                else:
                    vc, vs, vm = vc_n, vs_n, vm_n
            new_cache = (kc, k_full, ks, km, vc, v_full, vs, vm, kv_seq_len)
        else:
            w = torch.matmul(query_states, _rep(key_states, n_rep).transpose(2, 3)) / math.sqrt(self.head_dim)
            w = masked_softmax(w)
            attn_output = torch.matmul(w, _rep(value_states, n_rep))
            new_cache = self._prefill_cache_kivi(key_states, value_states) if use_cache else None
        if attn_output.size() != (bsz, self.num_heads, q_len, self.head_dim):
            raise ValueError(f"`attn_output` should be of size {(bsz, self.num_heads, q_len, self.head_dim)}, but is {attn_output.size()}")
        attn_output = attn_output.transpose(1, 2).contiguous().reshape(bsz, q_len, self.num_heads * self.head_dim)
        if self.tp_world > 1:
            from .parallel import all_gather_heads
            attn_output = all_gather_heads(attn_output, self.tp_world, self.tp_group)
        return self.o_proj(attn_output), None, (new_cache if use_cache else None)


class LlamaForCausalLM_KIVI(LlamaForCausalLM_GEARKIVI):
    """modeling_llama_kivi.py:516-: the causal LM over LlamaAttention_KIVI layers (same weights layout as the GEAR model)."""

    def __init__(self, config, compress_config=None):
        cc = dict(compress_config or {})
        cc.setdefault("residual", config.residual_length)
        cc.setdefault("compress_method", "KIVI")
        cc.setdefault("group_size", config.group_size)
        cc.setdefault("quantize_bit", config.k_bits)
        super().__init__(config, cc)
        for i, layer in enumerate(self.model.layers):
            old = layer.self_attn
            new = LlamaAttention_KIVI(i, config, cc)
            new.load_state_dict(old.state_dict())
            layer.self_attn = new.to(old.q_proj.weight.device, old.q_proj.weight.dtype)


@dataclass
class MistralConfigLite(LlamaConfigLite):
    """The MistralConfig attributes the attention reads (defaults: Mistral-7B: 32 query / 8 KV heads, theta 1e4, window 4096)."""
    intermediate_size: int = 14336
    num_key_value_heads: int = 8
    max_position_embeddings: int = 32768
    sliding_window: Optional[int] = 4096


def _check_window(config, max_tokens):
    sw = getattr(config, "sliding_window", None)
    if sw is not None and max_tokens > sw:
        raise NotImplementedError(f"contexts beyond the sliding window ({sw} tokens) are not built: within the window Mistral's "
                                  "attention is Llama's with grouped KV heads")


class MistralForCausalLM_GEAR(LlamaForCausalLM_GEARKIVI):
    """Mistral-shaped model (GQA) over the GEAR attention hook: GenerationBench/.../Simulated/modeling_mistral.py:679-762 applies
    the same compress hook to MistralAttention.  Contexts up to the sliding window."""

    @torch.no_grad()
    def generate(self, input_ids, max_length: int, use_cache: bool = True):
        _check_window(self.config, max_length)
        return super().generate(input_ids, max_length, use_cache)


class MistralForCausalLM_KIVI(LlamaForCausalLM_KIVI):
    @torch.no_grad()
    def generate(self, input_ids, max_length: int, use_cache: bool = True):
        _check_window(self.config, max_length)
        return super().generate(input_ids, max_length, use_cache)
