"""HBM rates torch's own kernels reach on this box (GPU box): fill (write only), copy (read + write), sum (read only), 1 GiB fp16."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.exp_rows import timeit  # noqa
n = 1 << 29
x = torch.randn(n, device="cuda", dtype=torch.float16)
y = torch.empty_like(x)
GiB = float(1 << 30)
t = timeit(lambda: y.fill_(1.0)); print(f"fill  1 GiB: {t:.3f} ms  {GiB / t / 1e9 * 1e3:.0f} GB/s written")
t = timeit(lambda: y.zero_()); print(f"zero  1 GiB: {t:.3f} ms  {GiB / t / 1e9 * 1e3:.0f} GB/s written")
t = timeit(lambda: y.copy_(x)); print(f"copy  1 GiB: {t:.3f} ms  {2 * GiB / t / 1e9 * 1e3:.0f} GB/s read + written")
xi = x.view(torch.int32)
t = timeit(lambda: xi.sum()); print(f"sum   1 GiB: {t:.3f} ms  {GiB / t / 1e9 * 1e3:.0f} GB/s read")
t = timeit(lambda: torch.add(x, 1.0, out=y)); print(f"add   1 GiB: {t:.3f} ms  {2 * GiB / t / 1e9 * 1e3:.0f} GB/s read + written")
