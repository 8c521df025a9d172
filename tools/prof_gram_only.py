"""PMC target (GPU box): the low-rank step on a token-major error tensor at bench size, wave-private Gram kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gear_amd import compress as C
E = (torch.randn(32, 32, 4096, 128, device="cuda", dtype=torch.float16) * 0.1).contiguous()
P0 = torch.rand(32, 32, 128, 8, device="cuda")
for _ in range(3):
    C.lowrank(E, 8, 3, P0)
torch.cuda.synchronize()
