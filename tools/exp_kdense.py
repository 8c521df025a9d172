"""A/B of the K chain's dense kernel at bench size: k_dense_kernel (default, csrc/kone.hip) against k_main_kernel (variant 128)."""
import sys, torch
sys.path.insert(0, ".")
from gear_amd import _lib as L, compress as C
lib = L.load()
def timed(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
torch.manual_seed(0)
x = torch.randn(32, 32, 4096, 128, dtype=torch.float16, device="cuda")
P0 = torch.rand(32, 32, 128, 8)
for bits, k, r, T in ((2, 40, 8, 4096), (4, 20, 4, 2048), (2, 0, 8, 4096), (2, 40, 0, 4096)):
    xx = x[:, :, :T].contiguous()
    PP = P0[..., :max(r, 1)].contiguous()
    for v in (128, 0, 128, 0):
        ms = timed(lambda: C.compress_key_fused(xx, bits, 64, k_out=k, rank=r, loop=3, mode="fp32", P0=PP if r else None, variant=v))
        ms_main = timed(lambda: C.compress_key_fused(xx, bits, 64, k_out=k, rank=r, loop=3, mode="fp32", P0=PP if r else None, variant=v | 8 | 16))
        print(f"bits {bits} k {k} r {r} T {T} variant {v:3d}: chain {ms:.4f} ms, dense kernel alone {ms_main:.4f} ms", flush=True)
    a = C.compress_key_fused(xx, bits, 64, k_out=k, rank=r, loop=3, mode="fp32", P0=PP if r else None, variant=128)
    b = C.compress_key_fused(xx, bits, 64, k_out=k, rank=r, loop=3, mode="fp32", P0=PP if r else None, variant=0)
    eq = [torch.equal(a.code, b.code), torch.equal(a.scale, b.scale), torch.equal(a.mn, b.mn)]
    if r:
        lra = torch.matmul(a.Q[:2].float(), a.P[:2].float().transpose(2, 3)); lrb = torch.matmul(b.Q[:2].float(), b.P[:2].float().transpose(2, 3))
        eq.append("lowrank rel %.2e" % float((lra - lrb).norm() / lra.norm()))
    print("equal:", eq, flush=True)
