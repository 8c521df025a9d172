"""Decode attention over the streaming cache (GPU box): outliers off / on through tiles / chunk tables / searched lists, vs the payload path.
One layer, 32 heads, prompt 4032 + 32 window tokens; per-call time from back-to-back launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gear_amd.cache import GearKVCache
from gear_amd import compress as C
from gear_amd.attention import decode_attention

H, D, T0 = 32, 128, 4032 + 32


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


torch.manual_seed(0)
k = torch.randn(1, H, T0, D).half().cuda()
v = torch.randn(1, H, T0, D).half().cuda()
q = torch.randn(1, H, 1, D).half().cuda()
for left in (0.0, 0.02):
    cc = dict(compress_method="gearslKIVI", group_size=64, residual=64, quantize_bit=2, rank=8, rankv=8, loop=3, left=left)
    c = GearKVCache(1, H, 4300, cc, "cuda")
    c.prefill(k, v)
    print(f"cache left={left}: {timed(lambda: c.attend(q)):.1f} us / call", flush=True)
    if left:
        c.use_tiles = False
        print(f"  lists + chunk tables:  {timed(lambda: c.attend(q)):.1f}", flush=True)
        c.use_chunk_index = False
        print(f"  lists, binary search:  {timed(lambda: c.attend(q)):.1f}", flush=True)
        c.use_tiles, c.use_chunk_index = True, True
# payload path (round-1 layout: lists of exactly k entries + chunk index)
kq, vq = k[:, :, :4032].contiguous(), v[:, :, :4032].contiguous()
for kk in (0, 40):
    pk = C.compress_key(kq, 2, 64, k_out=kk, rank=8, loop=3, mode="fp16")
    pv = C.compress_value(vq, 2, 64, k_out=kk, rank=8, loop=3, mode="fp16")
    print(f"payload k={kk}: {timed(lambda: decode_attention(q, pk, pv, k[:, :, 4032:], v[:, :, 4032:])):.1f} us / call", flush=True)
