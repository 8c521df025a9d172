#!/usr/bin/env python3
"""harness.py -- counterpart of the reference's timing harness cuda_supported_gear/test.py (:25-102): generate with a Llama-2-7B
shaped model through the GEAR cache, the KIVI cache or a plain fp16 cache and print wall time, peak GPU memory and tokens/s.

Same knobs as the reference (--batch_size, --model with "gearl" / "KIVI" / "None" in it; prompt 1000 tokens, generation up to
1500 -- test.py:25-28), plus what the reference hard-codes: --prompt_len, --max_length, the compress_config fields, --layers to
shrink the model, --all to run the three variants back to back.  There is no network here: weights are random (Llama shapes) and
the prompt is synthetic token ids, which does not change the timing of a greedy decode.

    python harness.py --model gearl --batch_size 8            # the reference's default run
    python harness.py --all --batch_size 1 --fast             # all variants; GEAR also through the fast decode path
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build(variant, mcfg, cc, dev):
    from gear_amd.modeling_llama_kivi import LlamaForCausalLM_KIVI
    from gear_amd.modeling_llamagear import LlamaAttention_GEAR, LlamaForCausalLM_GEARKIVI, _rep, apply_rotary_pos_emb
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float16)
    try:
        torch.manual_seed(0)
        with torch.device(dev):
            if variant == "gearl":
                return LlamaForCausalLM_GEARKIVI(mcfg, cc).eval()
            if variant == "KIVI":
                return LlamaForCausalLM_KIVI(mcfg, cc).eval()

            class LlamaAttention_FP16(LlamaAttention_GEAR):
                """The uncompressed baseline (transformers' LlamaForCausalLM in the reference, test.py:56-60): fp16 K / V cache."""

                def forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_value=None,
                            output_attentions=False, use_cache=False, **kw):
                    b, q, _ = hidden_states.shape
                    qs = self.q_proj(hidden_states).view(b, q, self.num_heads, self.head_dim).transpose(1, 2)
                    ks = self.k_proj(hidden_states).view(b, q, self.num_key_value_heads, self.head_dim).transpose(1, 2)
                    vs = self.v_proj(hidden_states).view(b, q, self.num_key_value_heads, self.head_dim).transpose(1, 2)
                    past = 0 if past_key_value is None else past_key_value[8]
                    if position_ids is None:
                        position_ids = torch.arange(past, past + q, device=hidden_states.device).unsqueeze(0)
                    cos, sin = self.rotary_emb(vs, position_ids)
                    qs, ks = apply_rotary_pos_emb(qs, ks, cos, sin)
                    if past_key_value is not None:
                        ks, vs = torch.cat([past_key_value[0], ks], 2), torch.cat([past_key_value[1], vs], 2)
                    n = self.num_key_value_groups
                    o = F.scaled_dot_product_attention(qs, _rep(ks, n), _rep(vs, n), is_causal=past_key_value is None and q > 1)
                    o = self.o_proj(o.transpose(1, 2).reshape(b, q, self.num_heads * self.head_dim))
                    return o, None, ((ks, vs, None, None, None, None, None, None, past + q) if use_cache else None)

            m = LlamaForCausalLM_GEARKIVI(mcfg, cc)
            for i, layer in enumerate(m.model.layers):
                a = LlamaAttention_FP16(i, mcfg, cc)
                a.load_state_dict(layer.self_attn.state_dict())
                layer.self_attn = a
            return m.eval()
    finally:
        torch.set_default_dtype(old)


def run(variant, args, dev):
    from gear_amd.modeling_llamagear import LlamaConfigLite
    mcfg = LlamaConfigLite(num_hidden_layers=args.layers, max_position_embeddings=max(4096, args.max_length + 64),
                           k_bits=args.bits, v_bits=args.bits, group_size=args.group_size, residual_length=args.residual)
    cc = dict(compress_method=args.compress_method, group_size=args.group_size, residual=args.residual, quantize_bit=args.bits,
              rank=args.rank, rankv=args.rank, loop=args.loop, left=args.left)
    model = build(variant, mcfg, cc, dev)
    torch.manual_seed(1)
    ids = torch.randint(0, mcfg.vocab_size, (args.batch_size, args.prompt_len), device=dev)
    torch.cuda.reset_peak_memory_stats(dev)
    torch.cuda.synchronize()
    t0 = time.time()
    if variant == "gearl" and args.fast:
        from gear_amd.fast_decode import FastGearDecoder
        out = FastGearDecoder(model, args.max_length + 8, batch=args.batch_size).generate(ids, args.max_length)
    elif variant == "None" and args.fast:
        # the control through the SAME decoder (fused GEMVs, gear_attn_decode_f16 over cache.Fp16KVCache): gearl and None then
        # differ in the cache only
        from gear_amd.fast_decode import FastGearDecoder
        out = FastGearDecoder(model, args.max_length + 8, batch=args.batch_size, cache_kind="fp16").generate(ids, args.max_length)
    else:
        out = model.generate(ids, args.max_length)
    torch.cuda.synchronize()
    dt = time.time() - t0
    peak = torch.cuda.max_memory_allocated(dev) / 2 ** 20
    new = (out.shape[1] - args.prompt_len) * args.batch_size
    print(f"[{variant}{' (fast path)' if variant in ('gearl', 'None') and args.fast else ''}] Peak memory usage on GPU: {peak:.1f} MB")
    print(f"[{variant}] time {dt:.3f}   ({new / dt:.1f} new tokens/s, batch {args.batch_size}, prompt {args.prompt_len} -> {out.shape[1]})")
    del model
    torch.cuda.empty_cache()
    return dict(variant=variant, fast=bool(args.fast and variant in ("gearl", "None")), batch_size=args.batch_size, prompt_len=args.prompt_len,
                max_length=int(out.shape[1]), time_s=dt, new_tokens_per_s=new / dt, peak_mem_MiB=peak, layers=args.layers)


def main():
    ap = argparse.ArgumentParser(description="GEAR / KIVI / fp16 generation timing (cuda_supported_gear/test.py)")
    ap.add_argument("--batch_size", type=int, default=8, help="Batch size.")                       # test.py:22
    ap.add_argument("--model", type=str, default="gearl", help='variant: contains "gearl", "KIVI" or "None"')   # test.py:23, :41-60
    ap.add_argument("--all", action="store_true", help="run gearl, KIVI and None back to back")
    ap.add_argument("--prompt_len", type=int, default=1000)                                         # max_token, test.py:26
    ap.add_argument("--max_length", type=int, default=1500)                                         # max_generation_length, :27
    ap.add_argument("--compress_method", default="gearlKIVI")                                       # test.py:31
    ap.add_argument("--group_size", type=int, default=64)
    ap.add_argument("--residual", type=int, default=64)
    ap.add_argument("--bits", type=int, default=2)
    ap.add_argument("--rank", type=int, default=2)
    ap.add_argument("--loop", type=int, default=3)
    ap.add_argument("--left", type=float, default=0.0, help="outlier fraction (fast path only: the hook's fused path stores none)")
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--fast", action="store_true", help="GEAR through FastGearDecoder (pre-allocated cache, fused kernels)")
    ap.add_argument("--json", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    variants = ["gearl", "KIVI", "None"] if args.all else [v for v in ("gearl", "KIVI", "None") if v in args.model]
    if not variants:
        raise SystemExit('--model must contain "gearl", "KIVI" or "None"')
    res = [run(v, args, dev) for v in variants]
    if args.json:
        print(json.dumps(res))


if __name__ == "__main__":
    main()
