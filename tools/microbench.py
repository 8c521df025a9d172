"""Kernel micro-benchmarks (GPU box): prints GB/s per kernel.  Not part of the test suite."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gear_amd.quant import new_pack, matmul


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    dev = "cuda:0"
    torch.manual_seed(0)
    B, H, T, D, g = 1, 32, 4096, 128, 64
    x = torch.randn(B, H, T, D, device=dev).half()
    xt = x.transpose(2, 3).contiguous()
    n = x.numel()
    print(torch.cuda.get_device_name(0))
    # copy baseline
    y = torch.empty_like(x)
    t = timeit(lambda: y.copy_(x))
    print(f"copy fp16 {n*2/2**20:.0f} MiB: {t*1e6:.1f} us  {n*4/t/1e12:.2f} TB/s (r+w)")
    for bits in (2, 4):
        t = timeit(lambda: new_pack.triton_quantize_and_pack_along_last_dim(x, g, bits))
        by = n * 2 + n * bits / 8 + 4 * n / g
        print(f"quant_lastdim V b{bits}: {t*1e6:.1f} us  {by/t/1e12:.2f} TB/s alg  ({n*2/t/1e9:.0f} GB/s fp16-in)")
        t = timeit(lambda: new_pack.triton_quantize_and_pack_along_last_dim_witherror(x, g, bits))
        by2 = by + n * 2
        print(f"quant_lastdim+err V b{bits}: {t*1e6:.1f} us  {by2/t/1e12:.2f} TB/s alg")
        t = timeit(lambda: new_pack.triton_quantize_and_pack_along_last_dim(xt, g, bits))
        print(f"quant_lastdim K^T b{bits}: {t*1e6:.1f} us  {by/t/1e12:.2f} TB/s alg")
        t = timeit(lambda: new_pack.quant_and_pack_kcache(x, g, bits))
        print(f"quant_k (token-major) b{bits}: {t*1e6:.1f} us  {by/t/1e12:.2f} TB/s alg")
        code, scale, mn = new_pack.triton_quantize_and_pack_along_last_dim(x, g, bits)
        t = timeit(lambda: new_pack.unpack_and_dequant_vcache(code, scale, mn, g, bits))
        print(f"dequant V b{bits}: {t*1e6:.1f} us  {by/t/1e12:.2f} TB/s alg")
        kc, ks, km = new_pack.triton_quantize_and_pack_along_last_dim(xt, g, bits)
        q = torch.randn(B, H, 1, D, device=dev).half()
        t = timeit(lambda: matmul.cuda_bmm_fA_qB_outer(g, q, kc, ks, km, bits))
        cb = n * bits / 8 + 4 * n / g
        print(f"gemv K-side b{bits}: {t*1e6:.1f} us  {cb/t/1e12:.2f} TB/s compressed  ({n*2/t/1e12:.1f} TB/s fp16-equiv)")
        a = torch.softmax(torch.randn(B, H, 1, T, device=dev), -1).half()
        t = timeit(lambda: matmul.cuda_bmm_fA_qB_outer(g, a, code, scale, mn, bits))
        print(f"gemv V-side b{bits}: {t*1e6:.1f} us  {cb/t/1e12:.2f} TB/s compressed")


if __name__ == "__main__":
    main()
