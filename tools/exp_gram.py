"""Gram kernel phase split (GPU box): lowrank on [32,32,S,128] for S = 64 (solve only), 1024, 4096; loop 1 vs 3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gear_amd import compress as C
from tools.exp_rows import timeit
for S in (64, 1024, 4096):
    E = (torch.randn(32, 32, S, 128, device="cuda") * 0.1).half()
    Et = E.transpose(2, 3).contiguous()
    P0 = torch.rand(32, 32, 128, 8, device="cuda")
    for loop in (1, 3):
        t = timeit(lambda: C.lowrank(E, 8, loop, P0))
        tt = timeit(lambda: C.lowrank(Et, 8, loop, P0, transposed=True))
        print(f"S={S:5d} loop={loop}: V layout {t:.3f} ms   K^T layout {tt:.3f} ms", flush=True)
