"""Chunk-size sweep for the cache-blocked compress (GPU box): ms per full K / V compress of the 7B / 4k cache."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gear_amd import compress as C
from tools.exp_rows import timeit  # noqa

L, H, T, D = 32, 32, 4096, 128
K = torch.randn(L, H, T, D, device="cuda", dtype=torch.float16)
V = torch.randn(L, H, T, D, device="cuda", dtype=torch.float16)
P0 = torch.rand(L, H, D, 8, device="cuda")
for mb in (0, 34, 68, 136, 272, 544):
    os.environ["GEAR_CHUNK_MB"] = str(mb)
    tk = timeit(lambda: C.compress_key(K, 2, 64, k_out=40, rank=8, loop=3, mode="fp32", P0=P0))
    tv = timeit(lambda: C.compress_value(V, 2, 64, k_out=40, rank=8, loop=3, mode="fp32", P0=P0))
    tk0 = timeit(lambda: C.compress_key(K, 2, 64, k_out=0, rank=8, loop=3, mode="fp16", P0=P0))
    print(f"chunk {mb:4d} MB: K (transpose+compress) {tk:.3f} ms   V {tv:.3f} ms   K fp16-mode no outliers {tk0:.3f} ms", flush=True)
