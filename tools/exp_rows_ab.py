"""A/B of the row compressor's experiment flags (GPU box): V layout, config 3 size, 2-bit, k = 40, error on."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gear_amd import compress as C, _lib as L


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


Ly, H, T, D = 32, 32, 4096, 128
x = torch.randn(Ly, H, T, D, device="cuda", dtype=torch.float16)
gv = (Ly * T, T, H * T * D, D, H, D, T * D)
flags = [0]
for rep in range(2):
    for f in flags:
        t = timeit(lambda: C.compress_rows_once(x, gv, 64, 2, 1, 40, True))
        t0 = timeit(lambda: C.compress_rows_once(x, gv, 64, 2, 1, 40, False))
        print(f"rows_exp={f}: k=40 err {t:.3f} ms   k=40 no err {t0:.3f} ms", flush=True)
