"""PMC target: the low-rank kernels at bench size (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gear_amd import compress as C
L, H, T, D = 32, 32, 4096, 128
E = (torch.randn(L, H, T, D, device="cuda", dtype=torch.float16) * 0.1).contiguous()
Et = C.transpose_last2(E)
P0 = torch.rand(L, H, D, 8, device="cuda")
for _ in range(3):
    C.lowrank(E, 8, 3, P0)
    C.lowrank(Et, 8, 3, P0, transposed=True)
torch.cuda.synchronize()
