"""rocprofv3 target: the K compress chain at bench size (32 layers x 32 heads x 4096 x 128, 2-bit, k = 40, rank 8) three times
through the kernel chain (kfused_one = -1) and three times with the single-read kernel (kfused_one = 1), after a 1 GiB copy as the
calibration of the PMC byte counters."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gear_amd import _lib as L, compress as C
lib = L.load()
torch.manual_seed(0)
x = torch.randn(32, 32, 4096, 128, dtype=torch.float16, device="cuda")
P0 = torch.rand(32, 32, 128, 8)
y = torch.empty_like(x)
for _ in range(3):
    y.copy_(x)
torch.cuda.synchronize()
del y
for one in (-1, 1):
    lib.gear_set_option(b"kfused_one", one)
    for _ in range(3):
        C.compress_key_fused(x, 2, 64, k_out=40, rank=8, loop=3, mode="fp32", P0=P0)
    torch.cuda.synchronize()
print("timeouts", lib.gear_kone_timeouts(), "fallback heads", lib.gear_kone_fallback_heads())
