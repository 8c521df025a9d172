"""PMC target (GPU box): the fp32 row compressor at config-3 size, V layout, 2 launches each of
(k=0, no err), (k=0, err), (k=40, no err), (k=40, err) -- in that order."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gear_amd import compress as C
Ly, H, T, D = 32, 32, 4096, 128
x = torch.randn(Ly, H, T, D, device="cuda", dtype=torch.float16)
gv = (Ly * T, T, H * T * D, D, H, D, T * D)
for k in (0, 40):
    for err in (False, True):
        for _ in range(2):
            C.compress_rows_once(x, gv, 64, 2, 1, k, err)
        torch.cuda.synchronize()
