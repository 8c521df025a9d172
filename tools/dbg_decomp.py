import sys, torch
sys.path.insert(0, ".")
from gear_amd import _lib as L, compress as C
lib = L.load()
torch.manual_seed(1)
for kind, shape, r in (("v", (1, 10, 256, 128), 8), ("v", (1, 32, 64, 128), 8), ("k", (1, 2, 1024, 128), 8), ("v", (1, 8, 64, 128), 4)):
    x = torch.randn(shape, dtype=torch.float16, device="cuda")
    P0 = torch.rand(shape[0], shape[1], 128, r, device="cuda")
    comp = C.compress_value if kind == "v" else C.compress_key
    p = comp(x, 2, 64, k_out=3, rank=r, loop=3, mode="fp32", P0=P0)
    lib.gear_set_option(b"decomp_general", 1)
    a = C.decompress(p, transposed_out=(kind == "k")).float()
    lib.gear_set_option(b"decomp_general", 0)
    b = C.decompress(p, transposed_out=(kind == "k")).float()
    d = (a - b).abs()
    bad = (d > 2e-3 * a.abs().clamp(min=1.0)) | torch.isnan(b)
    print(kind, shape, r, "bad", int(bad.sum()), "of", bad.numel(), "max", float(d[~torch.isnan(d)].max()))
    if bad.any():
        idx = bad.nonzero()
        print(" first bad idx", idx[:12].tolist())
        # per-dimension histograms
        for dim in range(4):
            u = idx[:, dim].unique()
            print("  dim", dim, "unique count", len(u), u[:40].tolist())
print("---- detail")
x = torch.randn((1, 2, 1024, 128), dtype=torch.float16, device="cuda")
P0 = torch.rand(1, 2, 128, 8, device="cuda")
p = C.compress_key(x, 2, 64, k_out=0, rank=8, loop=3, mode="fp32", P0=P0)
lib.gear_set_option(b"decomp_general", 1)
a = C.decompress(p, transposed_out=True).float()
lib.gear_set_option(b"decomp_general", 0)
b = C.decompress(p, transposed_out=True).float()
lr = torch.matmul(p.P.float(), p.Q.float().transpose(2, 3))   # [B,H,D,T]
base = a - lr
print("k=0: bad", int(((a - b).abs() > 2e-3).sum()))
for d in range(0, 8):
    print("row d=%d" % d, "general", [round(float(v), 3) for v in a[0, 0, d, :6]], "fast", [round(float(v), 3) for v in b[0, 0, d, :6]], "base", [round(float(v), 3) for v in base[0, 0, d, :6]], "lr", [round(float(v), 4) for v in lr[0, 0, d, :6]])
# which lr row does the fast path seem to add?
