"""PMC / kernel-trace target (GPU box): the compress and decompress kernels of one bench step at BASELINE config 3 size (config 2 with
GEAR_PROF_CONFIG=c2), 3 launches each, after a copy of the same tensor as calibration (bytes read = bytes written = the tensor).
usage: python tools/prof_step.py [layers]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gear_amd import compress as C

L = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H, T, D, g, bits, rank, k = 32, 4096, 128, 64, 2, 8, 40
if os.environ.get("GEAR_PROF_CONFIG") == "c2":          # BASELINE configs[1]: T = 2048, 4 bits, rank 4, 1 % outliers
    T, bits, rank, k = 2048, 4, 4, C.outlier_count(1, H, 2048, D, 0.01)
torch.manual_seed(0)
x = torch.empty((L, H, T, D), dtype=torch.float16, device="cuda")
for l in range(L):
    x[l] = torch.randn((H, T, D), device="cuda").half()
y = torch.empty_like(x)
for _ in range(3):
    y.copy_(x)
torch.cuda.synchronize()
del y
P0 = torch.rand((L, H, D, rank), device="cuda")
for _ in range(3):
    pv = C.compress_value(x, bits, g, k_out=k, rank=rank, loop=3, mode="fp32", P0=P0)
    if _ < 2:
        del pv
torch.cuda.synchronize()
for _ in range(3):                      # (round 4: the decompress kernels of the step too)
    yv = C.decompress(pv, transposed_out=True)
    del yv
torch.cuda.synchronize()
del pv
for path in ("fused", "rows"):
    for _ in range(3):
        pk = C.compress_key(x, bits, g, k_out=k, rank=rank, loop=3, mode="fp32", P0=P0, path=path)
        if not (path == "fused" and _ == 2):
            del pk
    torch.cuda.synchronize()
    if path == "fused":
        for _ in range(3):
            yk = C.decompress(pk, transposed_out=True)
            del yk
        torch.cuda.synchronize()
        del pk
