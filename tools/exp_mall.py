"""Does a weight matrix that was just read stay in the 256 MB Infinity Cache, and does a GEMV that hits there run faster?
GB/s of gear_gemv_f16 on ONE matrix repeated (cache-resident if it fits) vs rotating over > 1 GB of matrices."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gear_amd import _lib as L
lib = L.load()
for K, N in [(4096, 4096), (4096, 12288), (4096, 22016), (11008, 4096)]:
    Ws = [torch.randn(N, K, device="cuda", dtype=torch.float16) for _ in range(max(2, int(2e9 // (N * K * 2))))]
    x = torch.randn(1, K, device="cuda", dtype=torch.float16)
    y = torch.empty(1, N, device="cuda", dtype=torch.float16)
    res = []
    for mode in ("rotate", "same"):
        seq = Ws if mode == "rotate" else [Ws[0]] * len(Ws)
        for W in seq[:4]:
            lib.gear_gemv_f16(L.ptr(x), L.ptr(W), 1, K, N, L.ptr(y), L.stream_ptr())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            for W in seq:
                lib.gear_gemv_f16(L.ptr(x), L.ptr(W), 1, K, N, L.ptr(y), L.stream_ptr())
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (3 * len(seq))
        res.append(f"{mode} {N*K*2/us/1e3:6.0f} GB/s ({us:.1f} us)")
    print(f"{N}x{K} ({N*K*2/2**20:.0f} MiB):", "   ".join(res), flush=True)
