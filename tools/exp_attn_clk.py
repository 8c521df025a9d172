"""Phase clocks of the matrix-core attention kernel (library built with -DGEAR_ATTN_CLK: a temporary instrumentation, not in the
shipped build).  usage: python tools/exp_attn_clk.py
Build first:  touch gear_amd/csrc/attention.hip && make -C gear_amd/csrc EXTRA="-DGEAR_ATTN_CLK"   (then touch it again and
`python -c "from gear_amd import _lib; _lib.build()"` to return to the shipped library: the clocks are compiled out of it)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gear_amd import _lib as L
from gear_amd.cache import GearKVCache

dev = torch.device("cuda")
lib = L.load()
fn = lib.gear_debug_attn_clk
fn.argtypes = [ctypes.c_void_p]
fn.restype = ctypes.c_int


def case(B, Hq, Hkv, T, rank, s, label, bits=2):
    cc = dict(compress_method="gearslKIVI" if s > 0 else "gearlKIVI", group_size=64, residual=64, quantize_bit=bits, rank=rank, rankv=rank,
              loop=3, left=s)
    torch.manual_seed(B)
    k = torch.randn((B, Hkv, T - 32, 128), device=dev, dtype=torch.float16)
    v = torch.randn((B, Hkv, T - 32, 128), device=dev, dtype=torch.float16)
    c = GearKVCache(B, Hkv, T + 64, cc, dev, heads_total=Hkv)
    c.prefill(k, v)
    q = torch.randn((B, Hq, 1, 128), device=dev, dtype=torch.float16)
    L.set_option("attn_mfma", 1)
    for _ in range(5):
        c.attend(q)
    torch.cuda.synchronize()
    buf = np.zeros((8192, 8), dtype=np.uint64)
    assert fn(buf.ctypes.data) == 0
    L.set_option("attn_mfma", 0)
    nb = min(8192, B * Hkv * ((T - 32) // 128), int(os.environ.get('CLK_NB', '8192')))
    t = buf[:nb].astype(np.int64)
    d = np.diff(t[:, :7], axis=1)
    names = ["locate + issue loads", "K tile stores", "A rows+barrier", "MFMA scores..barrier", "softmax+V tile..barrier", "MFMA out..end"]
    print(f"{label}: {nb} workgroups; cycles per phase (mean / median / p90):")
    for i, n in enumerate(names):
        print(f"   {n:28s} {d[:, i].mean():8.0f} {np.median(d[:, i]):8.0f} {np.percentile(d[:, i], 90):8.0f}")
    top = t[:, 1] - t[:, 7]
    print(f"   last chunk: loop top -> phase 1 (locate, issue)  {top.mean():8.0f} {np.median(top):8.0f} {np.percentile(top, 90):8.0f}")
    last = t[:, 6] - t[:, 7]
    print(f"   last chunk: loop top -> end  {last.mean():8.0f}")
    life = t[:, 6] - t[:, 0]
    span = t[:, 6].max() - t[:, 0].min()
    print(f"   workgroup life mean {life.mean():.0f} cycles; kernel span {span} cycles; clock ticks ~ span/time")


case(16, 64, 8, 4096, 16, 0.0, "70B layer B=16 no outliers")
case(16, 64, 8, 4096, 16, 0.02, "70B layer B=16 2% outliers")
