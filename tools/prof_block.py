"""rocprofv3 target (GPU box): the decode-time block boundary at Llama-2-7B shapes -- the single-launch block compressor five
times, then the kernel chain it replaced five times (select, fused quantize + Gram, solve, Q pass, row compressor, Gram + solve,
Q pass, chunk index, two tile builders).  usage: python tools/prof_block.py [layers heads]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gear_amd import cache as gc

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H = int(sys.argv[2]) if len(sys.argv) > 2 else 32
cc = dict(compress_method="gearslKIVI", group_size=64, residual=64, quantize_bit=2, rank=8, rankv=8, loop=3, left=0.02)
for use_block in (True, False):
    gc.USE_BLOCK_KERNEL = use_block
    pool = gc.GearKVCachePool(layers, 1, H, 4096 + 256, cc, "cuda", seed=1)
    caches = [gc.GearKVCache(1, H, 4096 + 256, cc, "cuda", pool=pool, layer=l) for l in range(layers)]
    torch.manual_seed(0)
    pool.buf["kwin"].copy_(torch.randn_like(pool.buf["kwin"]))
    pool.buf["vwin"].copy_(torch.randn_like(pool.buf["vwin"]))
    for c in caches:
        c.seg0, c.kk0 = 4032, 40
    for _ in range(5):
        for c in caches:
            c.n_comp, c.n_win = 4032, 64
        pool.compress_all()
    torch.cuda.synchronize()
    del pool, caches
print("done")
