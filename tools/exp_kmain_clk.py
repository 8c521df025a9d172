"""Phase clocks of k_main_kernel (library built with GEAR_KF_CLK: temporary instrumentation).  usage: python tools/exp_kmain_clk.py
Build first:  touch gear_amd/csrc/kfused.hip && make -C gear_amd/csrc EXTRA="-DGEAR_KF_CLK=1"   (then touch it again and
`python -c "from gear_amd import _lib; _lib.build()"` to return to the shipped library: the clocks are compiled out of it)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gear_amd import _lib as L
from gear_amd import compress as C

lib = L.load()
fn = lib.gear_debug_kf_clk
fn.argtypes = [ctypes.c_void_p]
fn.restype = ctypes.c_int
Lr, H, T, D, g, bits, rank, k = 32, 32, 4096, 128, 64, 2, 8, 40
torch.manual_seed(0)
x = torch.empty((Lr, H, T, D), dtype=torch.float16, device="cuda")
for l in range(Lr):
    x[l] = torch.randn((H, T, D), device="cuda").half()
P0 = torch.rand((Lr, H, D, rank), device="cuda")
for _ in range(3):
    pk = C.compress_key(x, bits, g, k_out=k, rank=rank, loop=3, mode="fp32", P0=P0, path="fused")
    del pk
torch.cuda.synchronize()
buf = np.zeros((4096, 8), dtype=np.uint64)
assert fn(buf.ctypes.data) == 0
t = buf.astype(np.int64)
NCLK = int(os.environ.get("KF_NCLK", "8"))
t = t[(t[:, 0] > 0) & (t[:, NCLK - 1] > 0)][:, :NCLK]
d = np.diff(t, axis=1)
names = ["wait for x (forced vmcnt 0)", "quantize tile", "payload + E tile stores", "issue next loads", "barrier 1", "Gram MFMAs (4 tiles)", "barrier 2"]
if NCLK == 6:
    names = ["wait for the tile (forced vmcnt 0)", "E half 0 -> LDS", "MFMA + Q store half 0", "E half 1 -> LDS", "(next loads) MFMA + Q store half 1"]
print(f"{len(t)} workgroups, last round of each; cycles mean / median / p90")
for i, n in enumerate(names):
    print(f"  {n:32s} {d[:, i].mean():8.0f} {np.median(d[:, i]):8.0f} {np.percentile(d[:, i], 90):8.0f}")
print("  round total", (t[:, NCLK - 1] - t[:, 0]).mean())
