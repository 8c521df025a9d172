"""GEMV shape sweep on the GPU box: GEAR_GEMV_SK (K split) variants x Llama-2-7B projection shapes, GB/s of weight bytes."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch
    from gear_amd import _lib as L
    lib = L.load()
    shapes = [(4096, 12288), (4096, 4096), (4096, 22016), (11008, 4096), (4096, 32000)]
    out = []
    for K, N in shapes:
        Ws = [torch.randn(N, K, device="cuda", dtype=torch.float16) for _ in range(max(2, int(3e9 // (N * K * 2))))]
        x = torch.randn(1, K, device="cuda", dtype=torch.float16)
        y = torch.empty(1, N, device="cuda", dtype=torch.float16)
        for W in Ws[:2]:
            lib.gear_gemv_f16(L.ptr(x), L.ptr(W), 1, K, N, L.ptr(y), L.stream_ptr())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3
        e0.record()
        for _ in range(reps):
            for W in Ws:          # rotate through > L2 + MALL worth of weights
                lib.gear_gemv_f16(L.ptr(x), L.ptr(W), 1, K, N, L.ptr(y), L.stream_ptr())
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (reps * len(Ws))
        out.append(f"{N*K*2/us/1e3:6.0f}")
    print(sys.argv[1], " ".join(out), flush=True)
else:
    print("cfg   qkv    o     gu    down  lm_head   (GB/s)")
    for cfg in "1 2 4".split():
        env = dict(os.environ, GEAR_GEMV_SK=cfg)
        subprocess.run([sys.executable, __file__, cfg], env=env)
