"""Decode-attention microbench on the GPU box: GearKVCache (CSG layout, factor segments) at Llama-2-7B head shapes,
planned (128-token chunks) vs generic kernel, rank 8 vs quantization only.  us per layer call (partial + reduce)."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch
    from gear_amd.cache import GearKVCache
    for method, T0 in (("gearlKIVI", 4040), ("KIVI", 4040), ("gearlKIVI", 1000)):
        cc = dict(compress_method=method, group_size=64, residual=64, quantize_bit=2, rank=8, rankv=8, loop=3)
        caches = []
        for i in range(8):
            c = GearKVCache(1, 32, 4200, cc, "cuda", 128, seed=i)
            c.prefill(torch.randn(1, 32, T0, 128).half().cuda(), torch.randn(1, 32, T0, 128).half().cuda())
            for _ in range(70):   # cross one block boundary: a second factor segment appears
                c.append(torch.randn(1, 32, 1, 128).half().cuda(), torch.randn(1, 32, 1, 128).half().cuda())
                c.maybe_compress()
            caches.append(c)
        q = torch.randn(1, 32, 1, 128).half().cuda()
        for c in caches:
            c.attend(q)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()   # GPU time only: the python / ctypes launch path is ~30 us per call
        with torch.cuda.graph(g):
            for _ in range(4):
                for c in caches:
                    c.attend(q)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        print(sys.argv[1], method, T0, f"{e0.elapsed_time(e1) * 1e3 / 320:.1f} us", flush=True)
else:
    for name, env in (("planned", {}), ("generic", {"GEAR_ATTN_GENERIC": "1"})):
        subprocess.run([sys.executable, __file__, name], env=dict(os.environ, **env))
