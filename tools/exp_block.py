"""Block-boundary cost: the single-launch block compressor against the chain it replaces (7B shapes by default).
python tools/exp_block.py [layers heads batch]"""
import sys
import torch
sys.path.insert(0, ".")
from gear_amd import cache as gc

layers, H, B = (int(a) for a in (sys.argv[1:4] + ["32", "32", "1"][len(sys.argv) - 1:]))
variant = sys.argv[4] if len(sys.argv) > 4 else "full"
cc = dict(compress_method="gearslKIVI", group_size=64, residual=64, quantize_bit=2, rank=8, rankv=8, loop=3, left=0.02)
if variant in ("nooutlier", "dense"):
    cc["left"] = 0.0
if variant in ("nolowrank", "dense"):
    cc["compress_method"] = "KIVI"
print("variant", variant, cc)
res = {}
for use_block in (False, True):
    gc.USE_BLOCK_KERNEL = use_block
    pool = gc.GearKVCachePool(layers, B, H, 4096 + 256, cc, "cuda", seed=1)
    caches = [gc.GearKVCache(B, H, 4096 + 256, cc, "cuda", pool=pool, layer=l) for l in range(layers)]
    torch.manual_seed(0)
    pool.buf["kwin"].copy_(torch.randn_like(pool.buf["kwin"]))
    pool.buf["vwin"].copy_(torch.randn_like(pool.buf["vwin"]))
    for c in caches:
        c.seg0, c.kk0 = 4032, (40 if cc['left'] else 0)
    def once():
        for c in caches:
            c.n_comp, c.n_win = 4032, 64
        pool.compress_all()
    for _ in range(3):
        once()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    n = 20
    ev[0].record()
    for _ in range(n):
        once()
    ev[1].record()
    torch.cuda.synchronize()
    res[use_block] = ev[0].elapsed_time(ev[1]) / n * 1e3
    print(("block kernel" if use_block else "chain       "), f"{res[use_block]:8.1f} us per block boundary "
          f"({layers} layers x {H} heads x B={B})", flush=True)
nbytes = layers * B * H * 64 * 128 * 2 * 2
print(f"fp16 in: {nbytes / 1e6:.1f} MB -> block kernel {nbytes / res[True] / 1e6:.3f} TB/s of input; status {gc.block_kernel_status()}")
