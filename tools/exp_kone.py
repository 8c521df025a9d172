"""A/B of the single-read K kernel (csrc/kone.hip, option kfused_one = 1) against the select + main chain (-1): payload equality
and time.  usage: python tools/exp_kone.py [quick]"""
import sys
import time
import torch
sys.path.insert(0, ".")
from gear_amd import _lib as L, compress as C

lib = L.load()


def run(x, b, g, k, r, one, P0):
    lib.gear_set_option(b"kfused_one", one)
    p = C.compress_key_fused(x, b, g, k_out=k, rank=r, loop=3, mode="fp32", P0=P0)
    torch.cuda.synchronize()
    return p


def timed(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


shapes = [((1, 2, 256, 128), 2, 64, 3, 8), ((1, 4, 1024, 128), 2, 64, 5, 8), ((2, 3, 320, 128), 4, 32, 2, 4), ((1, 8, 4096, 128), 2, 64, 40, 8),
          ((1, 8, 2048, 128), 4, 64, 20, 4), ((1, 3, 8192, 128), 2, 64, 10, 16), ((1, 2, 1024, 128), 2, 64, 0, 8), ((1, 5, 4096, 128), 2, 32, 25, 8)]
if len(sys.argv) > 1 and sys.argv[1] == "big":
    shapes = []
for shape, b, g, k, r in shapes:
    torch.manual_seed(5)
    x = torch.randn(shape).half().cuda()
    P0 = torch.rand(shape[0], shape[1], 128, r)
    p0 = run(x, b, g, k, r, -1, P0)
    p1 = run(x, b, g, k, r, 1, P0)
    ok = {n: bool(torch.equal(getattr(p0, n), getattr(p1, n))) for n in ("code", "scale", "mn")}
    if k:
        ok["oidx"] = bool(torch.equal(p0.oidx, p1.oidx))
        ok["oval"] = bool(torch.equal(p0.oval.view(torch.int16), p1.oval.view(torch.int16)))
    lr0 = torch.matmul(p0.Q.float(), p0.P.float().transpose(2, 3))
    lr1 = torch.matmul(p1.Q.float(), p1.P.float().transpose(2, 3))
    rel = float((lr0 - lr1).norm() / lr0.norm())
    ncode = int((p0.code != p1.code).sum())
    print(shape, b, g, k, r, ok, "code words differing", ncode, "lowrank rel", "%.2e" % rel, "timeouts", lib.gear_kone_timeouts(), flush=True)

# bench size: 32 layers x 32 heads x 4096 x 128
torch.manual_seed(0)
x = torch.randn(32, 32, 4096, 128, dtype=torch.float16, device="cuda")
P0 = torch.rand(32, 32, 128, 8)
for one in (-1, 1, -1, 1):
    lib.gear_set_option(b"kfused_one", one)
    ms = timed(lambda: C.compress_key_fused(x, 2, 64, k_out=40, rank=8, loop=3, mode="fp32", P0=P0))
    ms_main = timed(lambda: C.compress_key_fused(x, 2, 64, k_out=40, rank=8, loop=3, mode="fp32", P0=P0, variant=16))
    print("kfused_one", one, "chain ms", round(ms, 4), "up to main ms", round(ms_main, 4), "timeouts", lib.gear_kone_timeouts(), flush=True)
p0 = run(x, 2, 64, 40, 8, -1, P0)
p1 = run(x, 2, 64, 40, 8, 1, P0)
print("big equal:", torch.equal(p0.code, p1.code), torch.equal(p0.scale, p1.scale), torch.equal(p0.mn, p1.mn), torch.equal(p0.oidx, p1.oidx),
      torch.equal(p0.oval.view(torch.int16), p1.oval.view(torch.int16)), "code words differing", int((p0.code != p1.code).sum()))
