"""K compress chain on the GPU box: the round-1 chain (transpose -> rows -> Gram -> Q pass) vs the fused token-major path
(select -> fused quantize + Gram -> solve -> Q pass), event-timed, BASELINE config 3 shapes by default.
usage: python tools/exp_kfused.py [layers] [T] [bits] [rank] [k]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gear_amd import compress as C

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
bits = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rank = int(sys.argv[4]) if len(sys.argv) > 4 else 8
k = int(sys.argv[5]) if len(sys.argv) > 5 else 40
H, D, g = 32, 128, 64
torch.manual_seed(0)
K = torch.empty((layers, H, T, D), dtype=torch.float16, device="cuda")
for l in range(layers):
    K[l] = torch.randn((H, T, D), device="cuda").half()
P0 = torch.rand((layers, H, D, rank), device="cuda")


def timed(fn, reps=5):
    for _ in range(2):
        out = fn()
        del out
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
        del out
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


n = K.numel()
only = os.environ.get("ONLY")
for name, fn in [
    ("rows  k+r", lambda: C.compress_key(K, bits, g, k_out=k, rank=rank, loop=3, mode="fp32", P0=P0, path="rows")),
    ("fused k+r", lambda: C.compress_key_fused(K, bits, g, k_out=k, rank=rank, loop=3, mode="fp32", P0=P0)),
    ("fused k+r generic tile", lambda: C.compress_key_fused(K, bits, g, k_out=k, rank=rank, loop=3, mode="fp32", P0=P0, variant=1)),
    ("fused k only", lambda: C.compress_key_fused(K, bits, g, k_out=k, rank=0, mode="fp32")),
    ("fused r only", lambda: C.compress_key_fused(K, bits, g, k_out=0, rank=rank, loop=3, mode="fp32", P0=P0)),
    ("fused quant only", lambda: C.compress_key_fused(K, bits, g, k_out=0, rank=0, mode="fp32")),
]+[("fused k+r no-tr", lambda: C.compress_key_fused(K, bits, g, k_out=k, rank=rank, loop=3, mode="fp32", P0=P0, variant=4))]:
    if only and only != name:
        continue
    try:
        ms = timed(fn)
        print(f"{name:26s} {ms:8.3f} ms   {2 * n / ms / 1e6:8.1f} GB/s of fp16 K", flush=True)
    except Exception as e:
        print(f"{name:26s} FAILED: {type(e).__name__}: {e}", flush=True)
