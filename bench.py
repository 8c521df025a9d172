#!/usr/bin/env python3
"""bench.py -- KV-cache compress + decompress throughput of the GEAR hot path on MI355X.

One "step" = one pass of the hot path over the whole KV cache of the named model at the named context:
    for every layer:  K, V fp16 [H, T, D]  --compress-->  packed payload (quantized backbone + rank-r factors +
    sparse outliers)  --decompress-->  fp16 K^T / V again
with the inputs resident in HBM.  value = fp16 KV bytes that went through compress PLUS through decompress,
per second, whole job (all ranks).

Workload (BASELINE.json): default = configs[2], the one the metric is quoted on
    "Llama-2-7B, seq=4096, 2-bit KIVI-style per-channel K / per-token V + rank-8 + 2% outlier, 1xMI355X"
    (--config c2 selects configs[1]: seq=2048, 4-bit, rank-4, 1%).
Multi-GPU: KV heads are sharded across ranks (7B: 32/N heads per GPU); compress / decompress need no data-path
collective (V outlier rows are selected per shard with k scaled by 1/N); total work is fixed ->
"scaling": "strong".

Launch:  python bench.py [--gpus N --steps K --warmup W]      (N > 1 via torch.distributed.run, one rank per GPU)
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (model, layers, kv_heads, head_dim, T, bits, group, rank, loop, sparsity)
    "c3": ("Llama-2-7B", 32, 32, 128, 4096, 2, 64, 8, 3, 0.02),
    "c2": ("Llama-2-7B", 32, 32, 128, 2048, 4, 64, 4, 3, 0.01),
}
HBM_PEAK_GBS = 8000.0  # MI355X spec (/opt/skills/guides/MI355X_MICROARCH.md)
TRAFFIC_C3 = 2.4917e9   # HBM bytes per compress_rows launch (mean of the V and K^T launches), profiles/r1_pmc_traffic_rows_fp32.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--layers", type=int, default=0, help="override the layer count (debug only; invalidates the number)")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="debug only: run ONE rank's shard of an N-GPU job on this GPU (H/N heads, k/N V outliers); invalidates the number")
    ap.add_argument("--streams", type=int, default=2, choices=(1, 2, 4, 8),
                    help="1: one stream; 2: the K chain and the V chain of a step run on two HIP streams (they are independent); "
                         "4 / 8: each of them additionally split into 2 / 4 groups of layers")
    ap.add_argument("--graph", action="store_true",
                    help="capture the step's launches once as a hipGraph and replay it (takes the host enqueue cost out of short "
                         "steps; measured within noise of eager launches at every size tried, so it is off by default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode", action="store_true", help="skip the full-model decode tokens/s leg")
    return ap.parse_args()


def cpu_baseline(cfg):
    """The oracle (CPU restatement of the reference's simulated path, validated against the reference in the build
    container) on a bounded sample of the same workload: ONE layer's K and V at full heads / context."""
    import numpy as np
    from oracle import oracle as orc
    model, layers, H, D, T, bits, group, rank, loop, s = cfg
    Hs, nl = H, 8  # bounded sample: 8 of the 32 layers, all heads, full context (about 10 s on the GPU box's host cores)
    rng = np.random.default_rng(0)
    k = rng.standard_normal((nl, Hs, T, D)).astype(np.float16)
    v = rng.standard_normal((nl, Hs, T, D)).astype(np.float16)
    P0k = rng.random((nl, Hs, D, rank), dtype=np.float32)
    P0v = rng.random((nl, Hs, D, rank), dtype=np.float32)
    orc.compress_insert_function(k[:1, :1, :256], v[:1, :1, :256], "GEAR", bits, group, rank, rank, loop, s,
                                 P0k[:1, :1], P0v[:1, :1])  # warm up / load the library
    t0 = time.perf_counter()
    orc.compress_insert_function(k, v, "GEAR", bits, group, rank, rank, loop, s, P0k, P0v)
    dt = time.perf_counter() - t0
    nbytes = 2 * (k.size + v.size) * 2  # quantize->dequantize round trip: counted like the GPU step
    return {
        "value": nbytes / dt / 1e9, "unit": "GB/s", "cores": orc.num_threads(), "kind": "port",
        "sample": f"oracle compress_insert_function(GEAR) on {nl} layers x {Hs} heads x T={T} (K+V), {dt:.2f} s",
    }


def decode_tokens_per_s(cfg, dev, new_tokens=64):
    """a14 counterpart (cuda_supported_gear/test.py:95-102): random-weight Llama-2-7B through the GEAR attention hook
    (packed cache, fused dequant GEMV, block compression every `residual` tokens), greedy decode, one synchronize
    before the clock stops.  Prefill = context - new_tokens so that decoding happens AT the named context."""
    import torch
    from gear_amd.modeling_llamagear import LlamaConfigLite, LlamaForCausalLM_GEARKIVI
    model_name, layers, H, D, T, bits, group, rank, loop, s = cfg
    mcfg = LlamaConfigLite(num_hidden_layers=layers, num_attention_heads=H, num_key_value_heads=H, hidden_size=H * D,
                           max_position_embeddings=max(4096, T), k_bits=bits, v_bits=bits, group_size=group, residual_length=64)
    cc = dict(compress_method="gearlKIVI", group_size=group, residual=64, quantize_bit=bits, rank=rank, rankv=rank, loop=loop)
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float16)
    try:
        with torch.device(dev):
            model = LlamaForCausalLM_GEARKIVI(mcfg, cc).eval()
    finally:
        torch.set_default_dtype(old)
    prompt = T - new_tokens
    ids = torch.randint(0, mcfg.vocab_size, (1, prompt), device=dev)
    res = {"context": T, "batch": 1, "weights": "random init, Llama-2-7B shapes",
           "method": "gearlKIVI %d-bit rank %d, residual 64 (CSG fused path)" % (bits, rank)}
    # (1) the build's fast path: pre-allocated GearKVCache + fused attention, ~10 launches per layer
    from gear_amd.fast_decode import FastGearDecoder
    torch.manual_seed(0)
    fast = FastGearDecoder(model, T + 2 * new_tokens + 8)
    nxt = fast.prefill(ids).argmax(-1, keepdim=True)
    for _ in range(2):
        nxt = fast.step(nxt).argmax(-1, keepdim=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(new_tokens - 2):
        nxt = fast.step(nxt).argmax(-1, keepdim=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res.update({"eager_fast_path_tokens_per_s": (new_tokens - 2) / dt})
    # (1b) the same token step captured once as a HIP graph (device-side pos / slot / T / W) and replayed
    n_graph = new_tokens - 2
    fast.tok.copy_(nxt)
    fast.step_graph()                                    # capture + first replay
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_graph):
        fast.step_graph()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    graph_tps = n_graph / dt
    eager_tps = res["eager_fast_path_tokens_per_s"]
    # headline = the faster of the two launch modes of the same token step (eager launches pipeline ahead of the GPU once the
    # step is down to ~200 kernels; graph replay wins when the host is slow)
    best = max(graph_tps, eager_tps)
    res.update({"tokens_per_s": best, "ms_per_token": 1e3 / best, "graph_replay_tokens_per_s": graph_tps,
                "path": "FastGearDecoder (GearKVCache + fused GEMVs + gear_attn_decode): "
                        + ("step_graph, one hipGraph per token step" if graph_tps >= eager_tps else "step, eager launches"),
                "peak_mem_MiB": torch.cuda.max_memory_allocated(dev) / 2 ** 20})
    del fast
    torch.cuda.empty_cache()
    # (2) the reference-shaped attention hook (17-slot tuple cache, torch.cat appends, ~60 eager ops per layer)
    torch.manual_seed(0)
    with torch.no_grad():
        logits, past = model(ids, None, True)
        nxt = logits[:, -1].argmax(-1, keepdim=True)
        for _ in range(2):
            logits, past = model(nxt, past, True)
            nxt = logits[:, -1].argmax(-1, keepdim=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(new_tokens - 2):
            logits, past = model(nxt, past, True)
            nxt = logits[:, -1].argmax(-1, keepdim=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    res["hook_module_tokens_per_s"] = (new_tokens - 2) / dt
    del model, past
    torch.cuda.empty_cache()
    return res


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # GEAR_BENCH_ONE_GPU=1 (debug only): every rank uses cuda:0 and the gloo backend, to exercise the multi-rank control
    # flow on a single-GPU box; the numbers it prints are meaningless
    one_gpu = world > 1 and os.environ.get("GEAR_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from gear_amd import compress as C
    from gear_amd import _lib
    _lib.load()

    cfg = CONFIGS[args.config]
    model, layers, H, D, T, bits, group, rnk, loop, sparsity = cfg
    if args.layers:
        layers = args.layers
    shards = args.emulate_world if (args.emulate_world and world == 1) else world
    assert H % shards == 0, "KV heads must divide across ranks"
    Hl = H // shards
    # k per side: reference formula on the FULL row (compress_function.py:300-303); per-shard V rows take k/N
    k_full = C.outlier_count(1, H, T, D, sparsity)
    k_key = k_full
    k_val = max(1, k_full // shards)

    torch.manual_seed(1234 + rank)
    K = torch.empty((layers, Hl, T, D), dtype=torch.float16, device=dev)
    V = torch.empty_like(K)
    for l in range(layers):  # fill in slices: no 4-byte temporaries of the whole cache
        K[l] = torch.randn((Hl, T, D), device=dev, dtype=torch.float32).half()
        V[l] = torch.randn((Hl, T, D), device=dev, dtype=torch.float32).half()
    P0k = torch.rand((layers, Hl, D, rnk), device=dev, dtype=torch.float32)
    P0v = torch.rand((layers, Hl, D, rnk), device=dev, dtype=torch.float32)

    ev = {}

    def stage(name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        ev.setdefault(name, []).append(e)

    def step_serial():
        stage("t0")
        # K^T re-layout (what the attention hook hands over, llamagear.py:268) + compress
        pk = C.compress_key(K, bits, group, k_out=k_key, rank=rnk, loop=loop, mode="fp32", P0=P0k)
        stage("k_compress")
        pv = C.compress_value(V, bits, group, k_out=k_val, rank=rnk, loop=loop, mode="fp32", P0=P0v)
        stage("v_compress")
        kr = C.decompress(pk, transposed_out=True)
        stage("k_decompress")
        vr = C.decompress(pv)
        stage("v_decompress")
        return pk, pv, kr, vr

    n_str = max(2, args.streams)
    pool = [torch.cuda.Stream(device=dev) for _ in range(n_str)]
    parts = n_str // 2                      # groups of layers per tensor kind
    lb = [(layers * i // parts, layers * (i + 1) // parts) for i in range(parts)]

    def step_streams():
        # K and V (and groups of layers) are independent: their compress -> decompress chains run on separate HIP streams,
        # so the VALU-bound row compressor of one chain overlaps the HBM-bound low-rank / decompress kernels of another
        cur = torch.cuda.current_stream()
        outs = []
        for i, (l0, l1) in enumerate(lb):
            sk, sv = pool[2 * i], pool[2 * i + 1]
            sk.wait_stream(cur)
            sv.wait_stream(cur)
            # (the V chain is enqueued first: it starts with the VALU-bound row compressor, which then overlaps the K chain's
            # HBM-bound re-layout; measured 1250 vs 1205 GB/s the other way round)
            with torch.cuda.stream(sv):
                pv = C.compress_value(V[l0:l1], bits, group, k_out=k_val, rank=rnk, loop=loop, mode="fp32", P0=P0v[l0:l1])
                vr = C.decompress(pv)
            with torch.cuda.stream(sk):
                pk = C.compress_key(K[l0:l1], bits, group, k_out=k_key, rank=rnk, loop=loop, mode="fp32", P0=P0k[l0:l1])
                kr = C.decompress(pk, transposed_out=True)
            outs.append((pk, pv, kr, vr))
        for st in pool:
            cur.wait_stream(st)
        return outs[0] if parts == 1 else outs

    step = step_streams if args.streams >= 2 else step_serial

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(max(1, args.warmup)):
        out = step()
    sync()
    run_step, graphed = step, False
    if args.graph:
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = step()
            graph.replay()                      # one untimed replay
            run_step, graphed = graph.replay, True
        except Exception as e:                  # capture is an optimisation of the host side only: fall back to eager
            print(f"[bench] graph capture failed ({type(e).__name__}: {e}); launching eagerly", file=sys.stderr)
            torch.cuda.synchronize()
    sync()
    ev.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step()
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    del out
    n_elem_rank = K.numel()                       # per tensor kind, this rank
    fp16_bytes_job = 2 * n_elem_rank * 2 * world  # K + V, whole job
    value = 2 * fp16_bytes_job * args.steps / dt / 1e9   # through compress + through decompress

    # ---- per-stage GPU time (HIP events on the launch stream): a dedicated serial pass, outside the timed region
    step_serial()                                # (untimed: first main-stream allocations of the 2.7 GB of outputs)
    torch.cuda.synchronize()
    ev.clear()
    pk = pv = kr = vr = out = None
    for _ in range(3):
        pk = pv = kr = vr = out = None           # (drop the previous outputs first: the allocator then reuses their blocks
        pk, pv, kr, vr = out = step_serial()     #  instead of a 60 ms first-touch allocation of a second 2.7 GB set)
    torch.cuda.synchronize()
    names = ["k_compress", "v_compress", "k_decompress", "v_decompress"]   # k_compress includes the K^T re-layout
    prev = "t0"
    stages = {}
    for nme in names:
        ms = [a.elapsed_time(b) for a, b in zip(ev[prev], ev[nme])]
        stages[nme] = sum(ms) / len(ms)
        prev = nme

    # ---- roofline of the dominant kernel: the V row compressor (outlier select + fill + quant + pack + error)
    # is timed as its own event interval in a dedicated loop (the compress stages above also contain the low-rank
    # launches).  Algorithmic bytes per launch (DESIGN.md "compress_rows"): read 2n; write codes n*b/8 +
    # scale/mn 8n/g (fp32) + error 2n + outliers rows*2k*4.
    # Launches are the ones the step issues: one per chunk of layers (gear_amd/compress.py cache blocking), V geometry
    # and K^T geometry alternating -- both are the same kernel symbol, so this is also what the rocprofv3 kernel
    # summary averages over.
    from gear_amd.compress import compress_rows_once, _batches_per_chunk
    nb = _batches_per_chunk(layers, Hl * T * D * 2)
    n = nb * Hl * T * D                               # elements per launch
    geom_v = (nb * T, T, Hl * T * D, D, Hl, D, T * D)
    geom_k = (nb * Hl * D, D, D * T, T, 1, T, 0)
    errb = torch.empty((nb, Hl, T, D), dtype=torch.float16, device=dev)
    ktb = C.transpose_last2(K[:nb])
    chunks = [(b0, b0 + nb) for b0 in range(0, layers - nb + 1, nb)]

    def rows_pass():
        for b0, b1 in chunks:
            compress_rows_once(V[b0:b1], geom_v, group, bits, 1, k_val, True, err=errb)
            compress_rows_once(ktb, geom_k, group, bits, 1, k_key, True, err=errb.view(nb, Hl, D, T))
    for _ in range(2):
        rows_pass()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        rows_pass()
    e1.record()
    torch.cuda.synchronize()
    rows_ms = e0.elapsed_time(e1) / (reps * 2 * len(chunks))
    k_avg = 0.5 * (nb * T * 2 * k_val + nb * Hl * D * 2 * k_key)   # outlier entries per launch (V rows / K^T rows)
    alg_bytes = 2 * n + n * bits / 8 + 8 * n / group + 2 * n + k_avg * 4
    achieved = alg_bytes / (rows_ms * 1e-3) / 1e9
    # HBM bytes per launch from the PMC passes committed in profiles/r1_pmc_traffic_rows_fp32.md (FETCH_SIZE x2 per the
    # gfx950 correction + WRITE_SIZE); measured for exactly these launches (C3, 1 GPU, all layers), null otherwise
    traffic = TRAFFIC_C3 if (args.config == "c3" and world == 1 and not args.layers and nb == layers) else None
    roofline = {"bound": "hbm", "kernel": f"compress_rows_fp32_kernel<{bits}, float> (V-layout and K^T-layout launches, {nb} layers each)",
                "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "alg_bytes_per_launch": alg_bytes, "ms_per_launch": rows_ms}

    # ---- decompress-into-attention: one decode token's attention over the compressed cache of ALL layers
    from gear_amd.attention import decode_attention
    from gear_amd import parallel
    qv = torch.randn((layers, Hl, 1, D), device=dev, dtype=torch.float16)

    def attn_step():
        o = decode_attention(qv, pk, pv)
        if dist is not None:       # head shards -> full [layers, 1, H*D] on every rank (latency-bound, a few KiB / layer)
            o = parallel.all_gather_heads(o.transpose(1, 2).reshape(layers, 1, Hl * D), world)
        return o

    for _ in range(3):
        attn_step()
    sync()
    a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    areps = 20
    a0.record()
    for _ in range(areps):
        attn_step()
    a1.record()
    sync()
    attn_ms = a0.elapsed_time(a1) / areps
    if dist is not None:
        t = torch.tensor([attn_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        attn_ms = float(t.item())
    payload_bytes_job = (pk.nbytes() + pv.nbytes()) * world

    if rank == 0:
        res = {
            "metric": "KV compress+decompress GB/s (fp16 KV bytes through compress + through decompress per second)",
            "value": value, "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32 arithmetic on fp16 data, int%d payload" % bits, "data": "synthetic",
            "config": {"workload": f"{model} KV cache, {layers} layers x {H} KV heads x T={T} x D={D}, "
                                   f"{bits}-bit g={group} per-channel K / per-token V + rank-{rnk} (loop {loop}) + "
                                   f"{sparsity * 100:.0f}% outliers (BASELINE configs[{2 if args.config == 'c3' else 1}])",
                       "parallelism": f"head-shard x{world}", "k_outliers_per_side": [k_key, k_val],
                       "streams": args.streams, "hipgraph": graphed},
            "compress_GBps": fp16_bytes_job / ((stages["k_compress"] + stages["v_compress"]) * 1e-3) / 1e9,
            "decompress_GBps": fp16_bytes_job / ((stages["k_decompress"] + stages["v_decompress"]) * 1e-3) / 1e9,
            "stage_ms": stages,
            "payload_ratio": (2 * n_elem_rank * 2) / (pk.nbytes() + pv.nbytes()),
            "attn_decode": {"ms_per_token_all_layers": attn_ms, "compressed_GBps": payload_bytes_job / (attn_ms * 1e-3) / 1e9,
                            "fp16_equiv_GBps": fp16_bytes_job / (attn_ms * 1e-3) / 1e9,
                            "collective": "all_gather of per-rank attention output" if world > 1 else None},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(cfg)
        if world == 1 and not args.no_decode and not args.layers:
            del K, V, kr, vr, pk, pv, out
            torch.cuda.empty_cache()
            res["decode"] = decode_tokens_per_s(cfg, dev)
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
