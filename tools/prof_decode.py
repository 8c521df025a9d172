"""Profile target: FastGearDecoder decode steps on Llama-2-7B shapes (GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gear_amd.modeling_llamagear import LlamaConfigLite, LlamaForCausalLM_GEARKIVI
from gear_amd.fast_decode import FastGearDecoder
dev = "cuda"
mcfg = LlamaConfigLite(k_bits=2, v_bits=2)
cc = dict(compress_method="gearslKIVI", group_size=64, residual=64, quantize_bit=2, rank=8, rankv=8, loop=3, left=float(os.environ.get("LEFT", "0.02")))
torch.set_default_dtype(torch.float16)
with torch.device(dev):
    model = LlamaForCausalLM_GEARKIVI(mcfg, cc).eval()
torch.set_default_dtype(torch.float32)
ids = torch.randint(0, 32000, (1, 4040), device=dev)
fast = FastGearDecoder(model, 4200)
nxt = fast.prefill(ids).argmax(-1, keepdim=True)
for _ in range(3):
    nxt = fast.step(nxt).argmax(-1, keepdim=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20
for _ in range(n):
    nxt = fast.step(nxt).argmax(-1, keepdim=True)
torch.cuda.synchronize()
print("ms/token", (time.perf_counter() - t0) / n * 1e3)
# host-only cost: same python path but measure without waiting (async queue depth) -> enqueue time
t0 = time.perf_counter()
for _ in range(n):
    nxt = fast.step(nxt)
    nxt = nxt.argmax(-1, keepdim=True)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("enqueue ms/token", (t1 - t0) / n * 1e3, "total", (time.perf_counter() - t0) / n * 1e3)
