import sys, time, torch
sys.path.insert(0, "/root/repo")
from gear_amd.modeling_llamagear import LlamaConfigLite, LlamaForCausalLM_GEARKIVI, LlamaAttention_GEAR
T, new = 4096, 64
mcfg = LlamaConfigLite(max_position_embeddings=T + 256)
cc = dict(compress_method="gearlKIVI", group_size=64, residual=64, quantize_bit=2, rank=8, rankv=8, loop=3)
torch.set_default_dtype(torch.float16)
torch.manual_seed(0)
with torch.device("cuda"):
    model = LlamaForCausalLM_GEARKIVI(mcfg, cc).eval()
torch.set_default_dtype(torch.float32)
ids = torch.randint(0, 32000, (1, T - new), device="cuda")
for fast in (True, False):
    LlamaAttention_GEAR.fast_decode = fast
    with torch.no_grad():
        logits, past = model(ids, None, True)
        nxt = logits[:, -1].argmax(-1, keepdim=True)
        for _ in range(2):
            logits, past = model(nxt, past, True)
            nxt = logits[:, -1].argmax(-1, keepdim=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(new - 2):
            logits, past = model(nxt, past, True)
            nxt = logits[:, -1].argmax(-1, keepdim=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("fast" if fast else "tuple", (new - 2) / dt, "tok/s", type(past[0]).__name__, flush=True)
    del past
