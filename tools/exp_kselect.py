"""k_select_kernel + k_select_fix_kernel alone at bench size (variant 32 of gear_compress_key_fused), HIP events."""
import sys, torch
sys.path.insert(0, ".")
from gear_amd import compress as C
def timed(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
torch.manual_seed(0)
x = torch.randn(32, 32, 4096, 128, dtype=torch.float16, device="cuda")
for bits, k, T in ((2, 40, 4096), (4, 20, 2048)):
    xx = x[:, :, :T].contiguous()
    for rep in range(2):
        print(f"k {k} T {T}: select + fix {timed(lambda: C.compress_key_fused(xx, bits, 64, k_out=k, rank=0, loop=3, mode='fp32', P0=None, variant=32)):.4f} ms", flush=True)
