"""Phase clocks of k_dense_kernel (library built with EXTRA=-DGEAR_KO_CLK): shader cycles per phase summed over the slabs of a workgroup
(thread 0 only), median over the workgroups of one launch at bench size.  usage: python tools/exp_kdense_clk.py"""
import ctypes, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from gear_amd import _lib as L, compress as C
lib = L.load()
torch.manual_seed(0)
x = torch.randn(32, 32, 4096, 128, dtype=torch.float16, device="cuda")
P0 = torch.rand(32, 32, 128, 8)
for r in (8, 0):
    for _ in range(3):
        C.compress_key_fused(x, 2, 64, k_out=40, rank=r, loop=3, mode="fp32", P0=P0 if r else None, variant=8 | 16)
    torch.cuda.synchronize()
    buf = np.zeros((16384, 16), np.uint64)
    lib.gear_debug_ko_clk.argtypes = [ctypes.c_void_p]
    rc = lib.gear_debug_ko_clk(buf.ctypes.data)
    t = buf[:2048].astype(np.float64)
    life = t[:, 7] - t[:, 0]
    print("rank", r, "rc", rc, "lifetime median %.0f p10 %.0f p90 %.0f cycles" % (np.median(life), np.percentile(life, 10), np.percentile(life, 90)))
    for i, n in enumerate(["wait DMA + barrier", "DMA issue + substitute + barrier", "dense (thread 0 wave)", "barrier after dense", "Gram"]):
        print("  %-32s median %8.0f  p90 %8.0f  (per slab %.0f)" % (n, np.median(t[:, 1 + i]), np.percentile(t[:, 1 + i], 90), np.median(t[:, 1 + i]) / 16))
    print("  launch span %.0f cycles" % (t[:, 7].max() - t[:, 0].min()))
