import os, sys, torch
sys.path.insert(0, ".")
from gear_amd import _lib as L, compress as C
lib = L.load()
def timed(f, n=5):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
torch.manual_seed(0)
x = torch.randn(32, 32, 4096, 128, dtype=torch.float16, device="cuda")
P0 = torch.rand(32, 32, 128, 8)
lib.gear_set_option(b"kfused_one", 1)
k = int(os.environ.get("KK", "40"))
r = int(os.environ.get("RR", "8"))
ms = timed(lambda: C.compress_key_fused(x, 2, 64, k_out=k, rank=r, loop=3, mode="fp32", P0=P0, variant=16))
print("fallback heads", lib.gear_kone_fallback_heads())
print("dbg", os.environ.get("GEAR_KONE_DBG"), "k", k, "r", r, "up to main ms", round(ms, 4), "timeouts", lib.gear_kone_timeouts(), flush=True)
