"""PMC target for HBM traffic of the dominant kernel: 3 launches of the V row compressor at bench size (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gear_amd import compress as C
L, H, T, D = 32, 32, 4096, 128
x = torch.randn(L, H, T, D, device="cuda", dtype=torch.float16)
gv = (L * T, T, H * T * D, D, H, D, T * D)
y = torch.empty_like(x)
for _ in range(3):
    y.copy_(x)           # calibration: 1 GiB read + 1 GiB written
torch.cuda.synchronize()
for _ in range(3):
    C.compress_rows_once(x, gv, 64, 2, 1, 40, True)
torch.cuda.synchronize()
