"""PMC target for HBM traffic of the dominant kernel: 3 launches each of the row compressor at bench size in the V layout
and in the K^T layout (GPU box), after a 1 GiB copy as calibration."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gear_amd import compress as C
L, H, T, D = 32, 32, 4096, 128
x = torch.randn(L, H, T, D, device="cuda", dtype=torch.float16)
gv = (L * T, T, H * T * D, D, H, D, T * D)
gk = (L * H * D, D, D * T, T, 1, T, 0)
y = torch.empty_like(x)
for _ in range(3):
    y.copy_(x)           # calibration: 1 GiB read + 1 GiB written
torch.cuda.synchronize()
err = torch.empty_like(x)
for _ in range(3):
    C.compress_rows_once(x, gv, 64, 2, 1, 40, True, err=err)
torch.cuda.synchronize()
xt = x.view(L, H, D, T)   # any fp16 data will do for the K^T geometry
for _ in range(3):
    C.compress_rows_once(xt, gk, 64, 2, 1, 40, True, err=err.view(L, H, D, T))
torch.cuda.synchronize()
