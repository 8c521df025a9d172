"""K chain by k_main slab count (option kfused_nslab): whole chain, select only, main only (HIP events, config 3 size)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gear_amd import _lib as L, compress as C
dev = torch.device("cuda")
layers, H, T, D = 32, 32, 4096, 128
K = torch.empty((layers, H, T, D), dtype=torch.float16, device=dev)
for l in range(layers):
    K[l] = torch.randn((H, T, D), device=dev).half()
P0 = torch.rand((layers, H, D, 8), device=dev)
def timed(fn, reps=8):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for ns in (0, 1, 2, 4):
    L.set_option("kfused_nslab", ns)
    full = timed(lambda: C.compress_key_fused(K, 2, 64, 40, 8, 3, "fp32", P0))
    main = timed(lambda: C.compress_key_fused(K, 2, 64, 40, 8, 3, "fp32", P0, variant=8 | 16))
    print(f"nslab {ns}: chain {full:.3f} ms  main {main:.3f} ms", flush=True)
