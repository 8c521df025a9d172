"""PMC / kernel-trace target (GPU box): one layer's decode attention over the streaming cache and over an fp16 cache of the same
length, Llama-2-7B shapes (32 heads, T = 4096, 2-bit, rank 8, 2 % outliers) at batch 16 and the 70B-shaped grouped-query layer (64 on
8 heads, rank 16) at batch 16, after a 1 GiB copy as calibration.  3 calls each.  usage: python tools/prof_attn.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gear_amd.cache import GearKVCache
from gear_amd.attention import decode_attention_f16

dev = torch.device("cuda")
x = torch.randn((1 << 29,), device=dev, dtype=torch.float16)
y = torch.empty_like(x)
for _ in range(3):
    y.copy_(x)
torch.cuda.synchronize()
del x, y


def case(B, Hq, Hkv, T, rank, s):
    cc = dict(compress_method="gearslKIVI" if s > 0 else "gearlKIVI", group_size=64, residual=64, quantize_bit=2, rank=rank, rankv=rank,
              loop=3, left=s)
    torch.manual_seed(B)
    k = torch.randn((B, Hkv, T - 32, 128), device=dev, dtype=torch.float16)
    v = torch.randn((B, Hkv, T - 32, 128), device=dev, dtype=torch.float16)
    c = GearKVCache(B, Hkv, T + 64, cc, dev, heads_total=Hkv)
    c.prefill(k, v)
    q = torch.randn((B, Hq, 1, 128), device=dev, dtype=torch.float16)
    for _ in range(3):
        c.attend(q)
    torch.cuda.synchronize()
    kf = torch.randn((B, Hkv, T, 128), device=dev, dtype=torch.float16)
    vf = torch.randn((B, Hkv, T, 128), device=dev, dtype=torch.float16)
    for _ in range(3):
        decode_attention_f16(q, kf, vf, T)
    torch.cuda.synchronize()


case(16, 32, 32, 4096, 8, 0.02)
case(16, 64, 8, 4096, 16, 0.02)
