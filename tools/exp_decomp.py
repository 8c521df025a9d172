"""Experiment: decompress_rows timing by payload flavour (GPU box)."""
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
    for (k, r) in ((40, 0), (0, 8), (40, 8)):
        p = comp(x, 2, 64, k_out=k, rank=r, loop=3, mode="fp32", P0=P0 if r else None)
        res = []
        for env in ({}, {"GEAR_DECOMP_RPB": "8"}, {"GEAR_DECOMP_RPB": "16", "GEAR_DECOMP_TROWS": "2"}, {"GEAR_DECOMP_RPB": "4"}):
            os.environ.pop("GEAR_DECOMP_RPB", None)
            os.environ.pop("GEAR_DECOMP_TROWS", None)
            os.environ.update(env)
            res.append(f"{timeit(lambda: C.decompress(p, transposed_out=True)):.3f}")
        print(f"decompress {kind} k={k:2d} r={r}: default (16 rows, table 4) / 8 rows / 16 rows table 2 / 4 rows = {' / '.join(res)} ms")
