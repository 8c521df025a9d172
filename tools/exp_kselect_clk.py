"""Phase clocks of k_select_kernel (library built with EXTRA=-DGEAR_KS_CLK): shader cycles between the stamps of thread 0, median over
the 4096 workgroups of one launch at bench size.  usage: python tools/exp_kselect_clk.py"""
import ctypes, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from gear_amd import _lib as L, compress as C
lib = L.load()
torch.manual_seed(0)
x = torch.randn(32, 32, 4096, 128, dtype=torch.float16, device="cuda")
for _ in range(3):
    C.compress_key_fused(x, 2, 64, k_out=40, rank=0, loop=3, mode="fp32", P0=None, variant=32)
torch.cuda.synchronize()
buf = np.zeros((4096, 8), np.uint64)
lib.gear_debug_ks_clk.argtypes = [ctypes.c_void_p]
print("rc", lib.gear_debug_ks_clk(buf.ctypes.data))
t = buf.astype(np.float64)
names = ["A: sample statistics + thresholds", "B: stream all tokens (wave 0)", "B: barrier", "row sums + mean", "C.1: thresholds of the lists (quad bisection)", "C.1: barrier", "C.2: outputs (8 channels per wave)"]
life = t[:, 7] - t[:, 0]
print("lifetime median %.0f p10 %.0f p90 %.0f" % (np.median(life), np.percentile(life, 10), np.percentile(life, 90)))
for i, n in enumerate(names):
    d = t[:, i + 1] - t[:, i]
    print("  %-48s median %8.0f  p90 %8.0f" % (n, np.median(d), np.percentile(d, 90)))
print("launch span %.0f cycles; workgroups resident at once ~ %.0f" % (t[:, 7].max() - t[:, 0].min(), life.sum() / (t[:, 7].max() - t[:, 0].min())))
