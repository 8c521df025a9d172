"""One-off check at bench size: the row decompressor's default route (128 rows per workgroup, matrix-core row term) against the general
loop -- equal to one unit in the last place -- for K^T and V of config 3."""
import sys, torch
sys.path.insert(0, ".")
from gear_amd import _lib as L, compress as C
lib = L.load()
torch.manual_seed(3)
x = torch.randn(32, 32, 4096, 128, dtype=torch.float16, device="cuda")
P0 = torch.rand(32, 32, 128, 8, device="cuda")
for kind in ("v", "k"):
    comp = C.compress_value if kind == "v" else C.compress_key
    p = comp(x, 2, 64, k_out=40, rank=8, loop=3, mode="fp32", P0=P0)
    a = C.decompress(p, transposed_out=(kind == "k"))
    lib.gear_set_option(b"decomp_general", 1)
    b = C.decompress(p, transposed_out=(kind == "k"))
    lib.gear_set_option(b"decomp_general", 0)
    worst = 0
    for l in range(32):
        ai, bi = a[l].view(torch.int16).to(torch.int32), b[l].view(torch.int16).to(torch.int32)
        oa = torch.where(ai < 0, -(ai & 0x7FFF), ai); ob = torch.where(bi < 0, -(bi & 0x7FFF), bi)
        worst = max(worst, int((oa - ob).abs().max()))
    print(kind, "max difference in units of the last place:", worst)
    assert worst <= 1
    del p, a, b
print("ok")
