"""Experiment: where does compress_rows spend its time?  (GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gear_amd import compress as C
from gear_amd.quant import new_pack


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    L, H, T, D = 32, 32, 4096, 128
    x = torch.randn(L, H, T, D, device="cuda", dtype=torch.float16)
    xt = C.transpose_last2(x)
    gv = (L * T, T, H * T * D, D, H, D, T * D)
    gk = (L * H * D, D, D * T, T, 1, T, 0)
    for bits in (2, 4):
        for k in (0, 40):
            for err in (False, True):
                t = timeit(lambda: C.compress_rows_once(x, gv, 64, bits, 1, k, err))
                print(f"rows V  b{bits} k={k:2d} err={err}: {t:.3f} ms")
        t = timeit(lambda: C.compress_rows_once(xt, gk, 64, bits, 1, 40, True))
        print(f"rows K^T b{bits} k=40 err=True: {t:.3f} ms")
        t = timeit(lambda: C.compress_rows_once(x, gv, 64, bits, 0, 0, True))
        print(f"rows V  b{bits} mode fp16 k=0 err=True: {t:.3f} ms")
        t = timeit(lambda: new_pack.triton_quantize_and_pack_along_last_dim(x, 64, bits))
        print(f"quant_lastdim b{bits} fp16: {t:.3f} ms")
        t = timeit(lambda: new_pack.triton_quantize_and_pack_along_last_dim_witherror(x, 64, bits))
        print(f"quant_lastdim+err b{bits} fp16: {t:.3f} ms")
        t = timeit(lambda: new_pack.triton_quantize_and_pack_along_last_dim(x, 64, bits, mode="fp32"))
        print(f"quant_lastdim b{bits} fp32: {t:.3f} ms")
    y = torch.empty_like(x)
    t = timeit(lambda: y.copy_(x))
    print(f"copy 1 GiB: {t:.3f} ms")
    P0 = torch.rand(L, H, D, 8, device="cuda")
    E = (x * 0.1).contiguous()
    t = timeit(lambda: C.lowrank(E, 8, 3, P0))
    print(f"lowrank V (gram) r8: {t:.3f} ms")
    Et = C.transpose_last2(E)
    t = timeit(lambda: C.lowrank(Et, 8, 3, P0, transposed=True))
    print(f"lowrank K^T (gram) r8: {t:.3f} ms")


if __name__ == '__main__':
    main()
