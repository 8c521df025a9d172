"""Low-rank step on a bench-size token-major error tensor (GPU box): Gram kernel flavours (option gram_fused: 1 = one kernel per
head with the solve inside, rounds 1-3; 2 = slab kernel with workgroup barriers + k_solve; 0 = wave-private slab kernel + k_solve)
and, for the wave-private kernel, the elimination builds (gram_nstg 5 = start positions rotated per head, 6 = no matrix-core work, 7 = loads only, 8 = no cross-wave
sum / output)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gear_amd import _lib as L
from gear_amd import compress as C
from tools.exp_rows import timeit
lib = L.load()
heads = int(sys.argv[1]) if len(sys.argv) > 1 else 32
E = (torch.randn(32, heads, 4096, 128, device="cuda") * 0.1).half()
P0 = torch.rand(32, heads, 128, 8, device="cuda")
for fused, nstg in ((1, 0), (2, 0), (0, 3), (0, 2), (0, 4), (0, 6), (0, 7), (0, 8), (0, 3), (1, 0)):
    lib.gear_set_option(b"gram_fused", fused)
    lib.gear_set_option(b"gram_nstg", nstg)
    t = timeit(lambda: C.lowrank(E, 8, 3, P0))
    print(f"heads/layer={heads} gram_fused={fused} gram_nstg={nstg}: lowrank {t:.3f} ms", flush=True)
lib.gear_set_option(b"gram_fused", 0)
lib.gear_set_option(b"gram_nstg", 0)
