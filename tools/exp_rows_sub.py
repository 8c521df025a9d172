"""A/B of the wave-per-row compressor's dense part at bench size: substituted row + dense16s (default) against outlier masks +
dense16 (option rows_masked), interleaved on one box; outputs compared bit for bit."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gear_amd import compress as C, _lib as L
lib = L.load()

def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

Ly, H, T, D = 32, 32, 4096, 128
torch.manual_seed(0)
x = torch.randn(Ly, H, T, D, device="cuda", dtype=torch.float16)
gv = (Ly * T, T, H * T * D, D, H, D, T * D)
err = torch.empty_like(x)
for bits in (2, 4):
    for k in (40, 20):
        for want_err in (True, False):
            res = {}
            for rep in range(2):
                for masked in (1, -1):
                    lib.gear_set_option(b"rows_masked", masked)
                    t = timeit(lambda: C.compress_rows_once(x, gv, 64, bits, 1, k, want_err, err))
                    res.setdefault(masked, []).append(t)
            outs = []
            for masked in (1, -1):
                lib.gear_set_option(b"rows_masked", masked)
                o = C.compress_rows_once(x, gv, 64, bits, 1, k, want_err, torch.empty_like(x) if want_err else None)
                torch.cuda.synchronize()
                outs.append(o)
            eq = [bool(torch.equal(a, b)) for a, b in zip(outs[0], outs[1]) if a is not None]
            print(f"b{bits} k={k} err={want_err}: masked {min(res[1]):.4f} ms, substituted {min(res[-1]):.4f} ms, equal {eq}", flush=True)
lib.gear_set_option(b"rows_masked", 0)
