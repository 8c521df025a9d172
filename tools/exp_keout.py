"""A/B of the fused K chain at bench size: k_dense_kernel writing the error matrix + the MFMA Q pass reading it (option kfused_eout)
against the recomputing Q pass (k_qpass_kernel); interleaved on one box, Q P^T compared."""
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
P0 = torch.rand(32, 32, 128, 16)
for bits, k, r, T in ((2, 40, 8, 4096), (4, 20, 4, 2048), (2, 40, 16, 4096)):
    xx = x[:, :, :T].contiguous()
    PP = P0[..., :r].contiguous()
    res = {}
    for rep in range(2):
        for eo in (0, 1):
            lib.gear_set_option(b"kfused_eout", eo)
            ms = timed(lambda: C.compress_key_fused(xx, bits, 64, k_out=k, rank=r, loop=3, mode="fp32", P0=PP))
            ms_d = timed(lambda: C.compress_key_fused(xx, bits, 64, k_out=k, rank=r, loop=3, mode="fp32", P0=PP, variant=8 | 16))
            res.setdefault(eo, []).append((ms, ms_d))
    out = []
    for eo in (0, 1):
        lib.gear_set_option(b"kfused_eout", eo)
        out.append(C.compress_key_fused(xx, bits, 64, k_out=k, rank=r, loop=3, mode="fp32", P0=PP))
    a, b = out
    eq = [torch.equal(a.code, b.code), torch.equal(a.scale, b.scale), torch.equal(a.mn, b.mn), torch.equal(a.P, b.P), torch.equal(a.Q, b.Q)]
    lra = torch.matmul(a.Q[:2].float(), a.P[:2].float().transpose(2, 3)); lrb = torch.matmul(b.Q[:2].float(), b.P[:2].float().transpose(2, 3))
    print(f"bits {bits} k {k} r {r} T {T}: recompute chain {min(v[0] for v in res[0]):.4f} ms (dense alone {min(v[1] for v in res[0]):.4f}), "
          f"eout chain {min(v[0] for v in res[1]):.4f} ms (dense alone {min(v[1] for v in res[1]):.4f}); equal code/scale/mn/P/Q {eq}, "
          f"lowrank rel diff {float((lra - lrb).norm() / lra.norm()):.2e}, Q max abs diff {float((a.Q.float() - b.Q.float()).abs().max()):.3e}", flush=True)
lib.gear_set_option(b"kfused_eout", 0)
