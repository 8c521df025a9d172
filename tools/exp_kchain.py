"""K chain per kernel on the GPU box (config-3 size): k_select / k_main / chain, event-timed through the stage entry points."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gear_amd import compress as C

L, H, T, D = 32, 32, 4096, 128
torch.manual_seed(0)
K = torch.randn(L, H, T, D, device="cuda").half()
P0 = torch.rand(L, H, D, 8, device="cuda")


def timed(fn, reps=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


full = timed(lambda: C.compress_key_fused(K, 2, 64, k_out=40, rank=8, loop=3, mode="fp32", P0=P0))
konly = timed(lambda: C.compress_key_fused(K, 2, 64, k_out=40, rank=0, mode="fp32"))
ronly = timed(lambda: C.compress_key_fused(K, 2, 64, k_out=0, rank=8, loop=3, mode="fp32", P0=P0))
qonly = timed(lambda: C.compress_key_fused(K, 2, 64, k_out=0, rank=0, mode="fp32"))
print(f"chain {full:.3f}  select+main(no lr) {konly:.3f}  main+solve+qpass(no select) {ronly:.3f}  quant only {qonly:.3f}  => select ~ {konly - qonly:.3f} ms", flush=True)
