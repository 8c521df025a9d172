"""The hook model's decode step: eager fused launches against the captured graph (LlamaModel_GEAR.graph_decode).  Logits must be
identical token by token (same kernels, same order); then tokens/s at Llama-2-7B shapes, 4k context, against FastGearDecoder."""
import sys, time
import torch
sys.path.insert(0, ".")
from gear_amd.modeling_llamagear import LlamaConfigLite, LlamaForCausalLM_GEARKIVI


def build(layers, hidden, heads, inter, vocab=32000):
    cfg = LlamaConfigLite(num_hidden_layers=layers, num_attention_heads=heads, num_key_value_heads=heads, hidden_size=hidden,
                          intermediate_size=inter, max_position_embeddings=8192, k_bits=2, v_bits=2, group_size=64, residual_length=64)
    cfg.vocab_size = vocab
    cc = dict(compress_method="gearlKIVI", group_size=64, residual=64, quantize_bit=2, rank=8, rankv=8, loop=3)
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float16)
    torch.manual_seed(0)
    with torch.device("cuda"):
        m = LlamaForCausalLM_GEARKIVI(cfg, cc).eval()
    torch.set_default_dtype(old)
    return m


def run(m, ids, n, graph, teacher=None):
    m.model.graph_decode = graph
    m.model._hook_graph = None
    outs, toks = [], []
    with torch.no_grad():
        torch.manual_seed(7)          # (the block compress draws its power-iteration bases from torch's generator)
        logits, past = m(ids, None, True)
        nxt = logits[:, -1].argmax(-1, keepdim=True)
        for i in range(n):
            if teacher is not None:
                nxt = teacher[i]
            toks.append(nxt)
            logits, past = m(nxt, past, True)
            outs.append(logits[:, -1].clone())
            nxt = logits[:, -1].argmax(-1, keepdim=True)
    return outs, toks


m = build(2, 512, 4, 1024, vocab=1000)
torch.manual_seed(1)
ids = torch.randint(0, 1000, (1, 200), device="cuda")
a, toks = run(m, ids, 150, False)
b, _ = run(m, ids, 150, True, teacher=toks)
bad = [i for i in range(150) if not torch.equal(a[i], b[i])]
print("steps", len(a), "differing steps", bad[:10], "max abs diff", max(float((x.float() - y.float()).abs().max()) for x, y in zip(a, b)))
ids2 = torch.randint(0, 1000, (2, 70), device="cuda")
a, toks = run(m, ids2, 80, False)
b, _ = run(m, ids2, 80, True, teacher=toks)
print("batch 2: differing steps", [i for i in range(80) if not torch.equal(a[i], b[i])][:10])
del m
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    sys.exit(0)
m = build(32, 4096, 32, 11008)
ids = torch.randint(0, 32000, (1, 4096 - 64), device="cuda")
from gear_amd.modeling_llamagear import LlamaDecoderLayer_GEAR
for graph, fold in ((False, False), (False, True), (True, True), (False, True)):
    LlamaDecoderLayer_GEAR.fold_norm_weights = fold
    m.model.graph_decode = graph
    m.model._hook_graph = None
    with torch.no_grad():
        logits, past = m(ids, None, True)
        nxt = logits[:, -1].argmax(-1, keepdim=True)
        for _ in range(4):
            logits, past = m(nxt, past, True)
            nxt = logits[:, -1].argmax(-1, keepdim=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(56):
            logits, past = m(nxt, past, True)
            nxt = logits[:, -1].argmax(-1, keepdim=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("graph" if graph else "eager", "fold" if fold else "own weights", "hook module tokens/s %.1f" % (56 / dt), flush=True)
    del past
from gear_amd.fast_decode import FastGearDecoder
fast = FastGearDecoder(m, 4096 + 200)
with torch.no_grad():
    logits = fast.prefill(ids)
    nxt = logits.argmax(-1, keepdim=True)
    for _ in range(4):
        nxt = fast.step(nxt).argmax(-1, keepdim=True)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(24):
            nxt = fast.step(nxt).argmax(-1, keepdim=True)
        torch.cuda.synchronize()
        print("FastGearDecoder eager tokens/s %.1f (no outliers in this config)" % (24 / (time.perf_counter() - t0)), flush=True)
