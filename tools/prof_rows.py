"""PMC target: a few launches of the row compressor (GPU box): default path, histogram path, k = 0."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gear_amd import compress as C
L, H, T, D = 8, 32, 4096, 128
x = torch.randn(L, H, T, D, device="cuda", dtype=torch.float16)
gv = (L * T, T, H * T * D, D, H, D, T * D)
for _ in range(3):
    C.compress_rows_once(x, gv, 64, 2, 1, 40, True)
torch.cuda.synchronize()
os.environ["GEAR_ROWS_HIST_ONLY"] = "1"
for _ in range(3):
    C.compress_rows_once(x, gv, 64, 2, 1, 40, True)
torch.cuda.synchronize()
del os.environ["GEAR_ROWS_HIST_ONLY"]
for _ in range(3):
    C.compress_rows_once(x, gv, 64, 2, 1, 0, True)
torch.cuda.synchronize()
