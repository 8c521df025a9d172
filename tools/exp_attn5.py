"""Round 5: one layer's decode attention over the streaming cache at batch 1 / 4 / 16 -- compressed (default mapping, and the
round-4 arrangements through the options) against the fp16-cache baseline -- and the 70B-shaped GQA head group (8 query heads per KV
head, T = 8192, rank 16) grouped vs one workgroup per query head.  usage: python tools/exp_attn5.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gear_amd import _lib as L
from gear_amd.cache import GearKVCache
from gear_amd.attention import decode_attention_f16

dev = torch.device("cuda")


def timed(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def case(B, Hq, Hkv, T, rank, s, label, bits=2):
    cc = dict(compress_method="gearslKIVI" if s > 0 else "gearlKIVI", group_size=64, residual=64, quantize_bit=bits, rank=rank, rankv=rank,
              loop=3, left=s)
    torch.manual_seed(B)
    k = torch.randn((B, Hkv, T - 32, 128), device=dev, dtype=torch.float16)
    v = torch.randn((B, Hkv, T - 32, 128), device=dev, dtype=torch.float16)
    c = GearKVCache(B, Hkv, T + 64, cc, dev, heads_total=Hkv)
    c.prefill(k, v)
    q = torch.randn((B, Hq, 1, 128), device=dev, dtype=torch.float16)
    res = {}
    for name, opts in (("default", {}), ("mfma", {"attn_mfma": 1}), ("vector", {"attn_mfma": -1})):
        for o, val in opts.items():
            L.set_option(o, val)
        res[name] = timed(lambda: c.attend(q))
        for o in opts:
            L.set_option(o, 0)
    kf = torch.randn((B, Hkv, T, 128), device=dev, dtype=torch.float16)
    vf = torch.randn((B, Hkv, T, 128), device=dev, dtype=torch.float16)
    res["fp16_cache"] = timed(lambda: decode_attention_f16(q, kf, vf, T))
    fp16_bytes = B * Hkv * T * 128 * 2 * 2
    print(f"{label} bits={bits} B={B} Hq={Hq} Hkv={Hkv} T={T} r={rank} s={s}: " + "  ".join(f"{n} {us:.1f} us" for n, us in res.items()) +
          f"   (fp16 cache {fp16_bytes / 2**20:.0f} MiB -> {fp16_bytes / res['fp16_cache'] / 1e3:.0f} GB/s)", flush=True)


for B in (1, 4, 16):
    case(B, 32, 32, 4096, 8, 0.02, "7B")
case(1, 32, 32, 4096, 8, 0.0, "7B no outliers")
for B in (1, 4, 16):
    case(B, 32, 32, 2048, 4, 0.01, "7B config 2", bits=4)
case(1, 8, 1, 8192, 16, 0.02, "70B shard (1 KV head)")
case(1, 64, 8, 8192, 16, 0.02, "70B whole layer")
case(4, 32, 8, 4096, 8, 0.0, "Mistral-like 4:1")
case(16, 32, 8, 4096, 8, 0.0, "Mistral-like 4:1")
case(64, 32, 8, 2048, 8, 0.0, "Mistral-like 4:1")
case(16, 64, 8, 4096, 16, 0.02, "70B whole layer")
case(32, 64, 8, 4096, 16, 0.0, "70B whole layer")
