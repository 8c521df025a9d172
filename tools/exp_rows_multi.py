"""Short token rows (head shards): the eight-rows-per-wave compressor (rows_multi.hip) against the workgroup kernel, and the low-rank
step on the same shapes.  usage: python tools/exp_rows_multi.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gear_amd import _lib as L, compress as C


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, (layers, H, T, k, r) in {"7B / 8 GPUs": (32, 4, 4096, 5, 8), "70B / 8 GPUs": (80, 1, 8192, 1, 16), "7B / 16 (2 heads)": (32, 2, 4096, 2, 8),
                                   "13B / 4 GPUs (1280: workgroup kernel)": (40, 10, 4096, 12, 8)}.items():
    torch.manual_seed(0)
    x = torch.randn(layers, H, T, 128, device="cuda").half()
    geom = (layers * T, T, H * T * 128, 128, H, 128, T * 128)
    err = torch.empty_like(x)
    res = {}
    for wg in (0, 1):
        L.set_option("rows_wg_only", wg)
        res[wg] = timed(lambda: C.compress_rows_once(x, geom, 64, 2, 1, k, True, err))
    L.set_option("rows_wg_only", 0)
    P0 = torch.rand(layers, H, 128, r, device="cuda")
    lr = timed(lambda: C.lowrank(err, r, 3, P0))
    gb = x.numel() * 2 / 1e9
    print(f"{name}: {layers * T} rows of {H * 128}, k = {k}: default {res[0]:.1f} us ({gb / res[0] * 1e6:.0f} GB/s of fp16 in), "
          f"workgroup kernel {res[1]:.1f} us; low-rank step (rank {r}) {lr:.1f} us")
