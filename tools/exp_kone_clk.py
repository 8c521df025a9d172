"""Phase clocks of k_one_kernel (library built with EXTRA=-DGEAR_KO_CLK): median shader cycles between the stamps, over the
workgroups of one launch at bench size.  usage: python tools/exp_kone_clk.py"""
import ctypes, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from gear_amd import _lib as L, compress as C
lib = L.load()
torch.manual_seed(0)
x = torch.randn(32, 32, 4096, 128, dtype=torch.float16, device="cuda")
P0 = torch.rand(32, 32, 128, 8)
lib.gear_set_option(b"kfused_one", 1)
for _ in range(3):
    C.compress_key_fused(x, 2, 64, k_out=40, rank=8, loop=3, mode="fp32", P0=P0, variant=16)
torch.cuda.synchronize()
buf = np.zeros((16384, 16), np.uint64)
lib.gear_debug_ko_clk.argtypes = [ctypes.c_void_p]
rc = lib.gear_debug_ko_clk(buf.ctypes.data)
t = buf.astype(np.float64)
names = ["slab loaded -> LDS", "stats MFMA + E0 publish", "E0 wait", "totals", "scan", "E1 publish", "E1 wait (counts)", "gather", "bisection",
         "E2 publish", "E2 wait", "thresholds load", "mark + lists", "dense", "Gram MFMA", "atomics"]
print("rc", rc, "workgroups", len(t))
life = t[:, 15] - t[:, 0]
print("lifetime median %.0f  p10 %.0f  p90 %.0f cycles" % (np.median(life), np.percentile(life, 10), np.percentile(life, 90)))
for i in range(15):
    d = t[:, i + 1] - t[:, i]
    print("%-26s median %8.0f  p90 %8.0f" % (names[i + 1] if False else ["load+LDS", "stats+E0 pub", "E0 wait", "totals", "scan", "E1 publish", "E1 wait", "gather", "bisection", "E2 publish", "E2 wait", "thr load + zero", "mark+lists+obits", "dense", "Gram MFMA", "atomics"][i], np.median(d), np.percentile(d, 90)))
span = (t[:, 15].max() - t[:, 0].min())
print("launch span %.0f cycles; sum of lifetimes / (512 slots) = %.0f" % (span, life.sum() / 512))
