"""Experiment: decompress_rows timing by payload flavour (GPU box), config-3 size."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gear_amd import compress as C
from tools.exp_rows import timeit  # noqa

L, H, T, D = 32, 32, 4096, 128
x = torch.randn(L, H, T, D, device="cuda", dtype=torch.float16)
P0 = torch.rand(L, H, D, 8, device="cuda")
for kind in ("v", "k"):
    comp = C.compress_value if kind == "v" else C.compress_key
    for (k, r) in ((0, 0), (40, 0), (0, 8), (40, 8)):
        p = comp(x, 2, 64, k_out=k, rank=r, loop=3, mode="fp32", P0=P0 if r else None)
        t = timeit(lambda: C.decompress(p, transposed_out=(kind == "k")))
        print(f"decompress {kind} k={k:2d} r={r}: {t:.3f} ms", flush=True)
        del p
