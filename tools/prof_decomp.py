"""PMC target (GPU box): decompress of V and K payloads at config-3 size, 3 launches each, flavours (k, r) = (0,0), (40,8)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gear_amd import compress as C
L, H, T, D = 32, 32, 4096, 128
x = torch.randn(L, H, T, D, device="cuda", dtype=torch.float16)
P0 = torch.rand(L, H, D, 8, device="cuda")
for kind in ("v", "k"):
    comp = C.compress_value if kind == "v" else C.compress_key
    for (k, r) in ((0, 0), (40, 8)):
        p = comp(x, 2, 64, k_out=k, rank=r, loop=3, mode="fp32", P0=P0 if r else None)
        for _ in range(3):
            y = C.decompress(p, transposed_out=True)
            del y
        torch.cuda.synchronize()
        del p
