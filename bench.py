#!/usr/bin/env python3
"""bench.py -- KV-cache compress + decompress throughput of the GEAR hot path on MI355X.

One "step" = one pass of the hot path over the whole KV cache of the named model at the named context:
    for every layer:  K, V fp16 [H, T, D]  --compress-->  packed payload (quantized backbone + rank-r factors +
    sparse outliers)  --decompress-->  fp16 K^T / V again
with the inputs resident in HBM.  value = fp16 KV bytes that went through compress PLUS through decompress,
per second, whole job (all ranks).

Workload (BASELINE.json): default = configs[2], the one the metric is quoted on
    "Llama-2-7B, seq=4096, 2-bit KIVI-style per-channel K / per-token V + rank-8 + 2% outlier, 1xMI355X"
    --config c2 = configs[1] (seq 2048, 4-bit, rank 4, 1 %), c4 = configs[3] (13B, 1 %), c5 = configs[4] (70B GQA, 8k, rank 16).
Multi-GPU: KV heads are sharded across ranks (7B: 32/N heads per GPU, 13B: 40/4, 70B: 8/8 = one KV head with its 8 query
heads); K compress and both decompressors need no data-path collective.  V outlier rows span ALL heads of a token
(compress_function.py:297-333): with a process group the shards run the EXACT cross-shard selection (one small all-gather of
candidates per call; the concatenated shard payloads are the unsharded payload bit for bit -- checked at start-up on a
256-token tensor, `shard_parity` in the line) and the k / N per-shard selection of rounds 1-3 is reported beside it
(`v_selection_per_shard`); `--emulate-world` has no process group and runs per-shard.  Total work is fixed -> "scaling":
"strong".  The decode leg shards the attention the same way and all-gathers the per-rank attention output (RCCL by default)
in front of the replicated o_proj.

Accounting: `roofline` = the compress CHAIN north_star names (all launches of one
gear_compress_key_fused / gear_compress_value_fused call, the lower of K and V), HIP events on the launch stream, ALGORITHMIC
bytes of SURVEY.md 8(d) verbatim (read 2n; write codes n b/8 + scale/mn 4n/g + factors + 6 bytes per outlier -- no error
term; what the build actually stores -- fp32 scale/mn in the simulated arithmetic, uint16 indices -- is `stored_bytes`);
`roofline.dominant_kernel` = the longest single kernel, timed alone AND (from the rocprofv3 trace of the bench step) inside the
two-stream step; `roofline_chain` = the same byte definition per whole stage; `kernels` = the large kernels one by one;
`traffic` comes from the newest profiles/r*_traffic.json that was measured on exactly the library that is loaded now.
`value` is wall time over steps whose K and V chains OVERLAP on two HIP streams; `stage_ms` / `roofline_chain` are measured in
a separate serial pass (one chain at a time), so their sum is larger than `ms_per_step`.

Launch:  python bench.py [--gpus N --steps K --warmup W]      (N > 1 via torch.distributed.run, one rank per GPU)
"""
import argparse
import hashlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    "c3": dict(model="Llama-2-7B", layers=32, q_heads=32, kv_heads=32, hidden=4096, inter=11008, T=4096, bits=2, group=64,
               rank=8, loop=3, s=0.02, idx=2),
    "c2": dict(model="Llama-2-7B", layers=32, q_heads=32, kv_heads=32, hidden=4096, inter=11008, T=2048, bits=4, group=64,
               rank=4, loop=3, s=0.01, idx=1),
    "c4": dict(model="Llama-2-13B", layers=40, q_heads=40, kv_heads=40, hidden=5120, inter=13824, T=4096, bits=2, group=64,
               rank=8, loop=3, s=0.01, idx=3),
    "c5": dict(model="Llama-2-70B (GQA)", layers=80, q_heads=64, kv_heads=8, hidden=8192, inter=28672, T=8192, bits=2, group=64,
               rank=16, loop=3, s=0.02, idx=4),
}
D = 128
HBM_PEAK_GBS = 8000.0  # MI355X spec (/opt/skills/guides/MI355X_MICROARCH.md); ~6300 GB/s is what a plain copy reaches


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--prewarm-s", type=float, default=4.0,
                    help="seconds of untimed steps before the warm-up steps (device clocks / power state); 0 = none")
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--layers", type=int, default=0, help="override the layer count (debug only; invalidates the number)")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="debug only: run ONE rank's shard of an N-GPU job on this GPU (H/N heads, k/N V outliers); invalidates the number")
    ap.add_argument("--streams", type=int, default=2, choices=(1, 2, 4, 8),
                    help="1: one stream; 2: the K chain and the V chain of a step run on two HIP streams (they are independent); "
                         "4 / 8: each of them additionally split into 2 / 4 groups of layers")
    ap.add_argument("--key-path", default="auto", choices=("auto", "fused", "rows"),
                    help="K compress route: fused token-major kernels, or the transpose + row-compressor chain")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode", action="store_true", help="skip the full-model decode tokens/s leg")
    ap.add_argument("--decode-tokens", type=int, default=64)
    ap.add_argument("--v-selection", default="exact", choices=("exact", "per_shard"),
                    help="sharded legs (--gpus N / --emulate-world N): V outliers selected over the FULL token row (one small "
                         "all-gather of candidates per compress call; the shard payloads are the unsharded payload) or k / N "
                         "inside the shard's own heads (rounds 1-3); the other one's V chain time is reported beside it")
    ap.add_argument("--exchange", default="both", choices=("peer", "collective", "both"),
                    help="world > 1, decode leg: the head-shard exchange -- 'peer' = stores into hipIpc-mapped peer memory "
                         "(gear_xchg_allgather), 'collective' = all_gather_into_tensor (RCCL), 'both' = one decode leg each")
    return ap.parse_args()


def lib_sha256():
    from gear_amd import _lib
    h = hashlib.sha256()
    with open(_lib.LIB_PATH, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def cpu_baseline(cfg):
    """The oracle (CPU restatement of the reference's simulated path, validated against the reference in the build container) on
    a bounded sample of the same workload: method GEAR on up to 8 layers' K and V at full heads / context.  Two forms, both
    reported: `value` = oracle.gear_tensor, the whole per-tensor round trip inside the C library (OpenMP over rows / heads,
    bit-identical to the glue form: tests/test_oracle_golden.py); `numpy_glue` = the function-by-function restatement the parity
    tests use, which spends most of its time in single-threaded numpy casts and transposes."""
    import numpy as np
    from oracle import oracle as orc
    H, T, bits, group, rank, loop, s = cfg["kv_heads"], cfg["T"], cfg["bits"], cfg["group"], cfg["rank"], cfg["loop"], cfg["s"]
    nl = max(1, min(8, (32 * 4096 * 8) // (H * T)))
    rng = np.random.default_rng(0)
    k = rng.standard_normal((nl, H, T, D)).astype(np.float16)
    v = rng.standard_normal((nl, H, T, D)).astype(np.float16)
    P0k = rng.random((nl, H, D, rank), dtype=np.float32)
    P0v = rng.random((nl, H, D, rank), dtype=np.float32)
    orc.gear_tensor(k[:1, :1, :256], "k", bits, group, s, rank, loop, P0k[:1, :1])      # warm up / load the library
    t0 = time.perf_counter()
    orc.gear_tensor(k, "k", bits, group, s, rank, loop, P0k)
    orc.gear_tensor(v, "v", bits, group, s, rank, loop, P0v)
    dt = time.perf_counter() - t0
    nbytes = 2 * (k.size + v.size) * 2  # quantize->dequantize round trip: counted like the GPU step
    t0 = time.perf_counter()
    orc.compress_insert_function(k[:1], v[:1], "GEAR", bits, group, rank, rank, loop, s, P0k[:1], P0v[:1])
    dt_glue = time.perf_counter() - t0
    return {
        "value": nbytes / dt / 1e9, "unit": "GB/s", "cores": orc.num_threads(), "kind": "port",
        "sample": f"oracle.gear_tensor (method GEAR: outliers + quantize + rank-{rank} power iteration, all in C / OpenMP) on "
                  f"{nl} layers x {H} heads x T={T}, K and V, {dt:.2f} s",
        "numpy_glue": {"value": nbytes / nl / dt_glue / 1e9, "unit": "GB/s",
                       "what": f"oracle.compress_insert_function(GEAR) on 1 layer, {dt_glue:.2f} s: the same arithmetic function by "
                               "function with numpy casts / transposes in between (single-threaded outside the C calls)"},
        # for scale: the reference's own torch CPU path, measured in the BUILD container (it cannot travel), BASELINE.md section 2
        "reference_torch_in_build_container": {"value": 0.15, "unit": "GB/s", "cores": 8,
                                               "what": "compress_insert_function(GEAR, prefill) 7B one layer T=4096, 8 vCPU Xeon 2.1 GHz; "
                                                       "oracle.gear_tensor on the same 8 vCPUs: 0.25 GB/s"},
    }


def block_boundary_cost(cfg, dev, Hl=None):
    """The decode-time block boundary (every 64 tokens the fp16 window of every layer is compressed in place behind the cache:
    cuda_supported_gear/modeling_llamagear.py:265-286, :335-378) at the configuration's shapes: the single-launch block compressor
    (csrc/block_fused.hip) against the kernel chain it replaced, HIP events around back-to-back boundaries on the launch stream.
    Algorithmic bytes per boundary (SURVEY 8d on n = layers x heads x 64 x 128 elements per tensor kind): read 2n, write codes +
    scale / mn (fp16) + factors + outlier lists."""
    from gear_amd import cache as gc
    H, T, bits, group, rnk, loop, s = (Hl or cfg["kv_heads"]), cfg["T"], cfg["bits"], cfg["group"], cfg["rank"], cfg["loop"], cfg["s"]
    layers = cfg["layers"]
    cc = dict(compress_method="gearslKIVI" if s > 0 else "gearlKIVI", group_size=group, residual=64, quantize_bit=bits, rank=rnk,
              rankv=rnk, loop=loop, left=s)
    out = {}
    tcap = min(T + 256, 16384)
    t_at = (min(T, tcap - 128) // 64) * 64 - 64
    for use_block in (True, False):
        gc.USE_BLOCK_KERNEL = use_block
        try:
            pool = gc.GearKVCachePool(layers, 1, H, tcap, cc, dev, seed=1, heads_total=cfg["kv_heads"])
            caches = [gc.GearKVCache(1, H, tcap, cc, dev, pool=pool, layer=l, heads_total=cfg["kv_heads"]) for l in range(layers)]
            torch.manual_seed(0)
            pool.buf["kwin"].copy_(torch.randn(pool.buf["kwin"].shape, device=dev, dtype=torch.float16))
            pool.buf["vwin"].copy_(torch.randn(pool.buf["vwin"].shape, device=dev, dtype=torch.float16))
            kk0 = min(pool.dims["kk0_max"], t_at // 2) if pool.dims["kk_blk"] else 0
            for c in caches:
                c.seg0, c.kk0 = t_at, kk0

            def once():
                for c in caches:
                    c.n_comp, c.n_win = t_at, 64
                pool.compress_all()
            for _ in range(3):
                once()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20
            e0.record()
            for _ in range(reps):
                once()
            e1.record()
            torch.cuda.synchronize()
            out["block_kernel_us" if use_block else "chain_us"] = e0.elapsed_time(e1) / reps * 1e3
            if use_block:
                d = pool.dims
                n = layers * H * 64 * D
                alg = 2 * (2 * n) + 2 * (n * bits / 8 + 4 * n / group) + 2 * 2 * rnk * (64 + D) * layers * H \
                    + layers * H * D * 2 * d["kk_blk"] * 4 + layers * 64 * 2 * d["kv"] * 4
                out["one_launch"] = bool(H >= gc.BLOCK_KERNEL_MIN_HEADS or H == 1 or not d["kv"])
                out["alg_bytes"] = alg
                out["outliers_per_side"] = {"k_block_row": d["kk_blk"], "v_row": d["kv"]}
            del pool, caches
        finally:
            gc.USE_BLOCK_KERNEL = True
        torch.cuda.empty_cache()
    out["achieved_GBps"] = out["alg_bytes"] / (out["block_kernel_us"] * 1e-6) / 1e9
    out["frac"] = out["achieved_GBps"] / HBM_PEAK_GBS
    out["shape"] = f"{layers} layers x {H} KV heads x 64 tokens x {D}, K and V, at context {t_at}"
    out["kernel"] = "block_compress_kernel (one launch: V row selection + K / V tiles + low-rank step)" if out["one_launch"] else \
        "kernel chain (2 - 3 KV heads per rank: the row duties would serialize; one head or four and more take the block kernel)"
    out["amortized_us_per_token"] = out["block_kernel_us"] / 64
    return out


def attn_decode_by_batch(cfg, dev):
    """One layer's decode attention over the streaming cache (GearKVCache: 2-bit codes + per-block factors + outlier tiles + fp16
    window) at batch 1, 4, 16: time per call and compressed bytes per second, beside the same attention over an UNCOMPRESSED fp16
    cache.  The calls ROTATE over several caches (32 at batch 1: the layers of the model) so that every call finds its cache in
    HBM as a decode step does -- the same cache called again and again sits in the 256 MB Infinity Cache, compressed or not
    (`us_per_call_warm`: that figure; rounds 1-5a printed it as us_per_call)."""
    from gear_amd.cache import GearKVCache
    from gear_amd.attention import decode_attention_f16
    H, Hq, T, bits, group, rnk, loop, s = (cfg["kv_heads"], cfg["q_heads"], cfg["T"], cfg["bits"], cfg["group"], cfg["rank"],
                                           cfg["loop"], cfg["s"])
    cc = dict(compress_method="gearslKIVI" if s > 0 else "gearlKIVI", group_size=group, residual=64, quantize_bit=bits,
              rank=rnk, rankv=rnk, loop=loop, left=s)
    out = {}

    def timed(fns, reps):
        for f in fns:
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            fns[i % len(fns)]()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    for B, ncache in ((1, 32), (4, 16), (16, 4)):
        torch.manual_seed(B)
        k = torch.randn((B, H, T - 32, D), device=dev, dtype=torch.float16)
        v = torch.randn((B, H, T - 32, D), device=dev, dtype=torch.float16)
        caches = []
        for i in range(ncache):
            c = GearKVCache(B, H, T + 64, cc, dev, seed=i)
            c.prefill(k, v)
            caches.append(c)
        c = caches[0]
        q = torch.randn((B, Hq, 1, D), device=dev, dtype=torch.float16)
        reps = 192
        us = timed([(lambda cc_=cc_: cc_.attend(q)) for cc_ in caches], reps)
        us_warm = timed([lambda: c.attend(q)], reps)
        nc = c.n_comp
        # bytes the kernels read per call: codes + scale / mn + token factors of the compressed tokens, the outlier tiles' entries
        nbytes = B * H * nc * D * bits / 8 * 2 + B * H * (nc // group) * D * 2 * 2 * 2 + B * H * nc * rnk * 2 * 2
        if s > 0:
            nbytes += int(c.kcnt[:, :, :(nc + 127) // 128].clamp(min=0).sum()) * 4 + int(c.vcnt[:, :, :nc // 64].clamp(min=0).sum()) * 4
        nbytes += B * H * c.n_win * D * 2 * 2
        out[f"B{B}"] = {"us_per_call": us, "us_per_call_warm": us_warm, "caches_rotated": ncache,
                        "compressed_GBps": nbytes / (us * 1e-6) / 1e9,
                        "fp16_equiv_GBps": B * H * (nc + c.n_win) * D * 2 * 2 / (us * 1e-6) / 1e9}
        # the UNCOMPRESSED baseline at the same shapes (the reference's harness times model "None" beside gearl / KIVI,
        # cuda_supported_gear/test.py:41-62): gear_attn_decode_f16 over an fp16 cache of the same length -- same split / merge
        # kernels, fp16 rows instead of the packed payload.  > 1: compression makes this batch's attention FASTER.
        Ttot = nc + c.n_win
        del c, caches, k, v
        torch.cuda.empty_cache()
        kf = torch.randn((B, H, Ttot, D), device=dev, dtype=torch.float16)
        vf = torch.randn((B, H, Ttot, D), device=dev, dtype=torch.float16)
        kvs = [(kf, vf)] + [(kf.clone(), vf.clone()) for _ in range(ncache - 1)]
        us16 = timed([(lambda a_=a_, b_=b_: decode_attention_f16(q, a_, b_)) for a_, b_ in kvs], reps)
        us16_warm = timed([lambda: decode_attention_f16(q, kf, vf)], reps)
        out[f"B{B}"].update({"fp16_cache_us_per_call": us16, "fp16_cache_us_per_call_warm": us16_warm,
                             "fp16_cache_GBps": B * H * Ttot * D * 2 * 2 / (us16 * 1e-6) / 1e9,
                             "speedup_vs_fp16_cache": us16 / us})
        del kf, vf, kvs, q
        torch.cuda.empty_cache()
    out["note"] = ("fp16_cache_*: gear_attn_decode_f16 over an UNCOMPRESSED fp16 cache of the same length (the reference harness's "
                   "model None); speedup_vs_fp16_cache < 1 means the compressed cache costs attention time at that batch.  us_per_call: "
                   "calls rotate over `caches_rotated` caches (every call finds its cache in HBM, as a decode step does); *_warm: the "
                   "same cache again and again (it then sits in the Infinity Cache)")
    return out


def decode_tokens_per_s(cfg, dev, world, rank, new_tokens=64, exchange="both"):
    """a14 counterpart (cuda_supported_gear/test.py:95-102): random-weight model of the named shapes through the GEAR cache
    (packed cache with per-block low-rank factors and sparse outliers, fused decode attention, block compression every 64
    tokens), greedy decode, one synchronize before the clock stops.  Prefill = context - new_tokens so that decoding happens
    AT the named context.  world > 1: head-sharded attention + all-gather of the attention output."""
    from gear_amd.modeling_llamagear import LlamaConfigLite, LlamaForCausalLM_GEARKIVI
    T, bits, group, rnk, loop, s = cfg["T"], cfg["bits"], cfg["group"], cfg["rank"], cfg["loop"], cfg["s"]
    mcfg = LlamaConfigLite(num_hidden_layers=cfg["layers"], num_attention_heads=cfg["q_heads"],
                           num_key_value_heads=cfg["kv_heads"], hidden_size=cfg["hidden"], intermediate_size=cfg["inter"],
                           max_position_embeddings=max(4096, T + 256), k_bits=bits, v_bits=bits, group_size=group,
                           residual_length=64)
    cc = dict(compress_method="gearslKIVI", group_size=group, residual=64, quantize_bit=bits, rank=rnk, rankv=rnk, loop=loop,
              left=s)
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float16)
    try:
        torch.manual_seed(0)                       # every rank holds the same (replicated) weights
        with torch.device(dev):
            model = LlamaForCausalLM_GEARKIVI(mcfg, cc).eval()
    finally:
        torch.set_default_dtype(old)
    prompt = T - new_tokens
    torch.manual_seed(1)
    ids = torch.randint(0, mcfg.vocab_size, (1, prompt), device=dev)
    res = {"context": T, "batch": 1, "weights": f"random init, {cfg['model']} shapes",
           "method": "gearslKIVI %d-bit, rank %d per block, %.0f%% outliers (V rows: reference count%s; K prompt rows: reference "
                     "count, K 64-token blocks: nominal count), residual 64"
                     % (bits, rnk, s * 100, ", selected over the full row across the ranks (FastGearDecoder v_selection='exact')" if world > 1 else ""),
           "parallelism": f"head-shard x{world}"}
    from gear_amd.fast_decode import FastGearDecoder

    def leg(mode):
        """One decode leg: eager token steps, then (when the exchange can be captured) the same step replayed as a hipGraph."""
        out = {}
        fast = FastGearDecoder(model, T + 2 * new_tokens + 8, tp_rank=rank, tp_world=world, tp_exchange=mode)
        capturable = fast.gather is None or fast.gather.capturable
        if world > 1:
            kind = type(fast.gather).__name__
            out["exchange"] = ("gear_xchg_allgather: stores into the peers' hipIpc-mapped memory, one launch per layer inside the "
                               "token-step graph" if kind == "PeerHeadGather" else
                               "all_gather_into_tensor of the attention output per layer (%s)%s"
                               % ("captured in the token-step graph" if capturable else "eager steps",
                                  "" if mode == "collective" else "; peer mapping failed: %s" % (fast.exchange_error,)))
            out["exchange_class"] = kind
        nxt = fast.prefill(ids).argmax(-1, keepdim=True)
        for _ in range(2):
            nxt = fast.step(nxt).argmax(-1, keepdim=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(new_tokens - 2):
            nxt = fast.step(nxt).argmax(-1, keepdim=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        c0 = fast.layers[0]["cache"]
        out.update({"eager_tokens_per_s": (new_tokens - 2) / dt,
                    "outliers_per_side": {"v_row": c0.kv, "k_prompt_row": c0.kk0, "k_block_row": c0.kk_blk}})
        best = out["eager_tokens_per_s"]
        if capturable:
            # the same token step captured once as a HIP graph (device-side pos / slot / T / W) and replayed
            n_graph = new_tokens - 2
            fast.tok.copy_(nxt)
            fast.step_graph()                                    # capture + first replay
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n_graph):
                fast.step_graph()
            torch.cuda.synchronize()
            out["graph_replay_tokens_per_s"] = n_graph / (time.perf_counter() - t0)
            best = max(best, out["graph_replay_tokens_per_s"])
        fast.check_exchange()
        fast.check_block_kernel()
        out["tokens_per_s"] = best
        fast.close()
        del fast
        torch.cuda.empty_cache()
        return out

    modes = ["peer"] if world == 1 else (["collective", "peer"] if exchange == "both" else [exchange])
    legs = {m: leg(m) for m in modes}
    head = legs[modes[0]]
    res.update({k: v for k, v in head.items() if k != "exchange_class"})
    if world > 1:
        # both exchanges of the head-sharded token step, each measured: the RCCL all-gather north_star names is the default and
        # the headline (round 5), the peer-store kernel the option
        res["exchange_modes"] = {m: {k: v for k, v in l.items() if k != "outliers_per_side"} for m, l in legs.items()}
        res["exchange_default"] = modes[0]
    best = head["tokens_per_s"]
    # headline = the faster launch mode of the same token step (both reported)
    res.update({"tokens_per_s": best, "ms_per_token": 1e3 / best,
                "path": "FastGearDecoder (GearKVCache with in-place block compress + fused GEMVs + gear_attn_decode_cache)",
                "peak_mem_MiB": torch.cuda.max_memory_allocated(dev) / 2 ** 20})
    if world == 1:
        # the reference harness's comparison (cuda_supported_gear/test.py:41-62 times the model "None" beside gearl / KIVI): the SAME
        # decoder over an UNCOMPRESSED fp16 cache (cache.Fp16KVCache + gear_attn_decode_f16), at batch 1 and at a serving batch --
        # tokens/s = batch x steps / time, eager steps on both sides
        def by_batch(kind, Bb, n_steps):
            dec = FastGearDecoder(model, T + n_steps + 72, batch=Bb, cache_kind=kind)
            idb = ids.expand(Bb, -1).contiguous()
            nx = dec.prefill(idb).argmax(-1, keepdim=True)
            for _ in range(2):
                nx = dec.step(nx).argmax(-1, keepdim=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n_steps):
                nx = dec.step(nx).argmax(-1, keepdim=True)
            torch.cuda.synchronize()
            dtb = time.perf_counter() - t0
            cache_bytes = (sum((lw["cache"].kwin.numel() + lw["cache"].vwin.numel()) * 2 for lw in dec.layers) if kind == "fp16" else
                           sum(t.numel() * t.element_size() for t in dec.pool.buf.values()))       # (allocated capacity, both sides)
            dec.close()
            del dec
            torch.cuda.empty_cache()
            return Bb * n_steps / dtb, cache_bytes
        cmp_b = {}
        for Bb in (1, 16):
            n_steps = 40 if Bb == 1 else 24                # (stays inside one 64-token block: no block boundary in the timed steps)
            g_tps, gb = by_batch("gear", Bb, n_steps)
            f_tps, fb = by_batch("fp16", Bb, n_steps)
            cmp_b["B%d" % Bb] = {"gear_tokens_per_s": g_tps, "fp16_cache_tokens_per_s": f_tps, "gear_vs_fp16_cache": g_tps / f_tps,
                                 "gear_cache_MiB": gb / 2 ** 20, "fp16_cache_MiB": fb / 2 ** 20}
        res["vs_fp16_cache"] = dict(cmp_b, note="same FastGearDecoder, weights and fused GEMVs; cache_kind 'gear' (compressed streaming "
                                    "cache) against 'fp16' (uncompressed, gear_attn_decode_f16); eager steps, prompt %d tokens; "
                                    "batch > 4 projects through the library GEMM on both sides; *_cache_MiB = ALLOCATED capacity of all layers (the "
                                    "compressed side includes its acceleration structures -- sparse tiles, chunk index -- and a channel-"
                                    "factor slot for every 64-token block of the capacity, used or not)" % prompt)
    if world == 1 and cfg["layers"] * cfg["hidden"] <= 32 * 4096:
        # the reference-shaped attention hook (17-slot tuple cache, torch.cat appends, ~60 eager ops per layer): what a
        # reference user gets from the documented import swap alone (no outliers: the reference's fused path stores none)
        cc2 = dict(cc, compress_method="gearlKIVI")
        for layer in model.model.layers:
            layer.self_attn.compress_config = cc2
        with torch.no_grad():
            logits, past = model(ids, None, True)
            nxt = logits[:, -1].argmax(-1, keepdim=True)
            for _ in range(2):
                logits, past = model(nxt, past, True)
                nxt = logits[:, -1].argmax(-1, keepdim=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n_plain = max(1, new_tokens - 2 - 8)          # (the window fills at the last timed step: these steps cross no block boundary)
            for _ in range(n_plain):
                logits, past = model(nxt, past, True)
                nxt = logits[:, -1].argmax(-1, keepdim=True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(new_tokens - 2 - n_plain):
                logits, past = model(nxt, past, True)
                nxt = logits[:, -1].argmax(-1, keepdim=True)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        res["hook_module_tokens_per_s_between_boundaries"] = n_plain / (t1 - t0)
        res["hook_module_note"] = ("the documented import swap (LlamaForCausalLM_GEARKIVI, reference-shaped attention hook): six fused launches "
                                   "per layer and token; hook_module_tokens_per_s includes one block boundary (every 64 tokens each layer "
                                   "compresses its window with the hook's own key_compression / value_compression -- the reference's "
                                   "operators and CPU-generator basis draws, layer by layer: ~0.4 ms per layer, where FastGearDecoder "
                                   "compresses all layers in one launch)")
        res["hook_module_tokens_per_s"] = (new_tokens - 2) / dt
        del past
    del model
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
    lib = _lib.load()

    cfg = dict(CONFIGS[args.config])
    if args.layers:
        cfg["layers"] = args.layers
    layers, H, T, bits, group, rnk, loop, sparsity = (cfg["layers"], cfg["kv_heads"], cfg["T"], cfg["bits"], cfg["group"],
                                                      cfg["rank"], cfg["loop"], cfg["s"])
    shards = args.emulate_world if (args.emulate_world and world == 1) else world
    assert H % shards == 0, "KV heads must divide across ranks"
    Hl = H // shards
    # k per side: reference formula on the FULL row (compress_function.py:265-267 / :300-303); K rows live inside a head
    # (count independent of the sharding).  V rows span the heads: sharded legs run the EXACT cross-shard selection (the full row's
    # k, a shard keeps what falls into its heads: csrc/vsel.hip); --v-selection per_shard = k / N inside the shard (rounds 1-3)
    k_full = C.outlier_count(1, H, T, D, sparsity)
    k_key = min(k_full, T // 2)
    v_exact = shards > 1 and k_full > 0 and args.v_selection == "exact"
    k_val = k_full if (v_exact or shards == 1) else (max(1, k_full // shards) if k_full else 0)
    k_val_per_shard = max(1, k_full // shards) if k_full else 0
    key_path = args.key_path

    torch.manual_seed(1234 + rank)
    K = torch.empty((layers, Hl, T, D), dtype=torch.float16, device=dev)
    V = torch.empty_like(K)
    for l in range(layers):  # fill in slices: no 4-byte temporaries of the whole cache
        K[l] = torch.randn((Hl, T, D), device=dev, dtype=torch.float32).half()
        V[l] = torch.randn((Hl, T, D), device=dev, dtype=torch.float32).half()
    P0k = torch.rand((layers, Hl, D, rnk), device=dev, dtype=torch.float32)
    P0v = torch.rand((layers, Hl, D, rnk), device=dev, dtype=torch.float32)

    def comp_k(x, p0):
        return C.compress_key(x, bits, group, k_out=k_key, rank=rnk, loop=loop, mode="fp32", P0=p0, path=key_path)

    # exact selection: with a process group the candidates of the other ranks arrive by all-gather; the one-GPU emulation of a
    # rank's shard holds the other ranks' candidates fixed (made once, outside the timed region, from random heads of the same
    # shape) and does this rank's share per step: its own candidates, the thresholds over all of them, the sharded compress
    emu_others = {}

    def others_for(nl):
        if nl not in emu_others:
            from gear_amd import parallel as P_
            g_ = torch.Generator(device=dev).manual_seed(99)
            oth = []
            for r in range(1, shards):
                vv = torch.randn((nl, Hl, T, D), device=dev, dtype=torch.float16, generator=g_)
                oth.append(P_.v_candidates(vv, k_val, r))
                del vv
            emu_others[nl] = torch.stack(oth)
        return emu_others[nl]

    def comp_v(x, p0):
        if not v_exact:
            return C.compress_value(x, bits, group, k_out=k_val, rank=rnk, loop=loop, mode="fp32", P0=p0)
        if world > 1:
            return C.compress_value(x, bits, group, k_out=k_val, rank=rnk, loop=loop, mode="fp32", P0=p0, shard=(rank, world, None))
        from gear_amd import parallel as P_
        own = P_.v_candidates(x.contiguous(), k_val, 0)
        thr_fill = P_.v_thresholds(torch.cat([own[None], others_for(x.shape[0])]), k_val, H * D, 1)
        return C.compress_value(x, bits, group, k_out=k_val, rank=rnk, loop=loop, mode="fp32", P0=p0, shard=(0, shards, thr_fill))

    def comp_v_per_shard(x, p0):
        return C.compress_value(x, bits, group, k_out=k_val_per_shard, rank=rnk, loop=loop, mode="fp32", P0=p0)

    ev = {}

    def stage(name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        ev.setdefault(name, []).append(e)

    def step_serial():
        stage("t0")
        pk = comp_k(K, P0k)
        stage("k_compress")
        pv = comp_v(V, P0v)
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
        # so the instruction-bound kernels of one chain overlap the HBM-bound kernels of the other
        cur = torch.cuda.current_stream()
        outs = []
        for i, (l0, l1) in enumerate(lb):
            sk, sv = pool[2 * i], pool[2 * i + 1]
            sk.wait_stream(cur)
            sv.wait_stream(cur)
            with torch.cuda.stream(sv):
                pv = comp_v(V[l0:l1], P0v[l0:l1])
                vr = C.decompress(pv)
            with torch.cuda.stream(sk):
                pk = comp_k(K[l0:l1], P0k[l0:l1])
                kr = C.decompress(pk, transposed_out=True)
            outs.append((pk, pv, kr, vr))
        for st in pool:
            cur.wait_stream(st)
        return outs[0] if parts == 1 else outs

    step = step_streams if args.streams >= 2 else step_serial
    comp_v_exact = comp_v

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # untimed: bring the device to its sustained state first, then the W warm-up steps of the contract.  The first process on a
    # fresh box measured the serial K chain at 1.46-1.61 ms for its first ~3 s of GPU work and the next process on the same box at
    # 1.30-1.34 ms (three boxes); with 5 s of steps in front the first process reads 1.30-1.32 ms too.
    prewarm_steps, prewarm_t0 = 0, time.perf_counter()
    if args.prewarm_s > 0:
        out = step()
        sync()
        t1 = time.perf_counter()
        out = step()
        sync()
        d1 = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(d1, op=dist.ReduceOp.MAX)        # (the same number of steps on every rank)
        n_pre = int(min(5000.0, args.prewarm_s / max(float(d1.item()), 1e-4)))
        for _ in range(n_pre):
            out = step()
        sync()
        prewarm_steps = n_pre + 2
    prewarm_seconds = time.perf_counter() - prewarm_t0
    for _ in range(max(1, args.warmup)):
        out = step()
    sync()
    ev.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # sharded legs: the same timed loop once more with the OTHER V selection (k / N inside the shard's own heads, rounds 1-3) --
    # `value` is the exact selection (the algorithm of N = 1), `value_per_shard_selection` stands beside it
    dt_other = None
    if v_exact:
        comp_v = comp_v_per_shard
        for _ in range(max(1, args.warmup)):
            out = step()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        sync()
        dt_other = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt_other], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_other = float(t.item())
        comp_v = comp_v_exact
    del out
    n = K.numel()                                 # elements per tensor kind, this rank
    BH = layers * Hl
    fp16_bytes_job = 2 * n * 2 * world            # K + V, whole job
    value = 2 * fp16_bytes_job * args.steps / dt / 1e9   # through compress + through decompress

    # ---- per-stage GPU time (HIP events on the launch stream): a dedicated serial pass, outside the timed region
    step_serial()                                # (untimed: first main-stream allocations of the outputs)
    torch.cuda.synchronize()
    ev.clear()
    pk = pv = kr = vr = out = None
    for _ in range(5):
        pk = pv = kr = vr = out = None           # (drop the previous outputs first: the allocator then reuses their blocks)
        pk, pv, kr, vr = out = step_serial()
    torch.cuda.synchronize()
    names = ["k_compress", "v_compress", "k_decompress", "v_decompress"]
    prev = "t0"
    stages = {}
    for nme in names:
        ms = [a.elapsed_time(b) for a, b in zip(ev[prev], ev[nme])]
        if os.environ.get("GEAR_BENCH_DEBUG"):
            print(nme, [round(v, 3) for v in ms], file=sys.stderr)
        stages[nme] = sorted(ms)[len(ms) // 2]      # median of the five passes (one allocator hiccup in three passes once read 9.6 ms into a mean)
        prev = nme

    # ---- algorithmic bytes (one K or V tensor of n elements on this rank): read 2n, write the payload
    # SURVEY.md 8(d) VERBATIM: codes n b/8 + scale/mn 4n/g (fp16 each) + factors 2r(T+D) per head + 6 bytes per outlier (fp16
    # value + uint32 index).  What the build stores differs by ~1 %: in the simulated (fp32) arithmetic scale / mn are float32
    # (8n/g) and the indices uint16 (4 bytes per outlier) -- `stored_bytes`, not used for any fraction.
    def payload_bytes(kind, stored=False):
        rows = BH * D if kind == "k" else layers * T
        # (a head shard under the exact selection keeps on average k / N of a row's outliers per side: that, not its list capacity)
        kk = k_key if kind == "k" else (k_val_per_shard if shards > 1 else k_val)
        return (n * bits / 8 + (8 if stored else 4) * n / group
                + 2 * rnk * (T + D) * BH                     # P, Q fp16
                + rows * 2 * kk * (4 if stored else 6))

    alg = {"k_compress": 2 * n + payload_bytes("k"), "v_compress": 2 * n + payload_bytes("v"),
           "k_decompress": payload_bytes("k") + 2 * n, "v_decompress": payload_bytes("v") + 2 * n}
    stored = {"k_compress": 2 * n + payload_bytes("k", True), "v_compress": 2 * n + payload_bytes("v", True),
              "k_decompress": payload_bytes("k", True) + 2 * n, "v_decompress": payload_bytes("v", True) + 2 * n}
    chain = {nme: {"alg_bytes": alg[nme], "stored_bytes": stored[nme], "ms": stages[nme],
                   "achieved": alg[nme] / (stages[nme] * 1e-3) / 1e9,
                   "frac": alg[nme] / (stages[nme] * 1e-3) / 1e9 / HBM_PEAK_GBS} for nme in names}

    # ---- sharded legs: the other V selection's chain time, and the start-up self-check "concatenated shard payloads == unsharded
    # payload" on a 256-token tensor (all ranks hold the same seeded full tensor; each compresses its head shard with the exact
    # selection in the cache's fp16-stepwise arithmetic, rank 0 also the whole tensor; shards are gathered and compared bit for bit)
    shard_info = None
    if shards > 1 and k_full > 0:
        def time_v(fn):
            for _ in range(2):
                fn(V, P0v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(3):
                fn(V, P0v)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / 3
        other = comp_v_per_shard if v_exact else None
        shard_info = {"v_selection": args.v_selection, "v_outliers_per_side": k_val,
                      "v_compress_ms": stages["v_compress"],
                      "per_shard": {"v_outliers_per_side": k_val_per_shard,
                                    "v_compress_ms": time_v(other) if other else stages["v_compress"]},
                      "launches_of_the_selection": "gear_vsel_candidates + all-gather + gear_vsel_thresholds; the row compressor takes "
                                                   "the selection (gear_compress_value_sharded)" if v_exact else "inside the row compressor"}
        from gear_amd import parallel as P_
        torch.manual_seed(4242)
        Tt, kt = 256, max(1, C.outlier_count(1, H, 256, D, sparsity))
        vt = torch.randn((2, H, Tt, D), device=dev, dtype=torch.float16)
        p0t = torch.rand((2, H, D, rnk), device=dev, dtype=torch.float32)
        full = C.compress_value(vt, bits, group, k_out=kt, rank=rnk, loop=loop, mode="fp16", P0=p0t)
        my = rank if world > 1 else 0
        if world > 1:
            mine = C.compress_value(vt[:, my * Hl:(my + 1) * Hl].contiguous(), bits, group, k_out=kt, rank=rnk, loop=loop, mode="fp16",
                                    P0=p0t[:, my * Hl:(my + 1) * Hl].contiguous(), shard=(my, world, None))
            parts = {n: P_.all_gather_stack(getattr(mine, n), world) for n in ("code", "scale", "mn")}
            cat = {n: torch.cat(list(t), 1) for n, t in parts.items()}
        else:
            sh_ = [vt[:, r * Hl:(r + 1) * Hl].contiguous() for r in range(shards)]
            thr_fill = P_.v_thresholds(torch.stack([P_.v_candidates(x_, kt, r) for r, x_ in enumerate(sh_)]), kt, H * D, 0)
            ps = [C.compress_value(x_, bits, group, k_out=kt, rank=rnk, loop=loop, mode="fp16",
                                   P0=p0t[:, r * Hl:(r + 1) * Hl].contiguous(), shard=(r, shards, thr_fill)) for r, x_ in enumerate(sh_)]
            cat = {n: torch.cat([getattr(p_, n) for p_ in ps], 1) for n in ("code", "scale", "mn")}
        shard_info["shard_parity"] = bool(all(torch.equal(cat[n], getattr(full, n)) for n in ("code", "scale", "mn")))
        shard_info["shard_parity_what"] = (f"{'ranks' if world > 1 else 'emulated head shards on one GPU'}: codes + scale + zero point of the "
                                           f"{shards} shards of a [2, {H}, 256, 128] tensor, concatenated, == the unsharded payload "
                                           "(fp16-stepwise arithmetic, exact selection)")
        del vt, full, cat

    # ---- the large kernels one by one (HIP events around back-to-back launches of ONE kernel on the launch stream)
    def timed(fn, reps=5, warm=2):
        for _ in range(warm):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    kernels = []
    # (1) the V row compressor: outlier select + fill + quantize + pack + error, one launch over all layers
    geom_v = (layers * T, T, Hl * T * D, D, Hl, D, T * D)
    errb = torch.empty((layers, Hl, T, D), dtype=torch.float16, device=dev)
    k_rows = k_val_per_shard if v_exact else k_val       # (the standalone row compressor selects for itself: the shard's own count)
    rows_out = C._alloc_rows(tuple(V.shape), layers * T, group, bits, 1, k_rows, dev)
    ms_rows = timed(lambda: C._compress_rows(V, geom_v, group, bits, 1, k_rows, rows_out, errb))
    b_rows = 2 * n + n * bits / 8 + 4 * n / group + layers * T * 2 * k_rows * 6      # SURVEY 8(d) verbatim
    rows_len = Hl * D
    rows_name = (f"compress_rows_wave_kernel<{bits}, {rows_len // 1024}, fast + fallback pass>" if rows_len % 1024 == 0 and (rows_len <= 5120 or rows_len == 8192) and k_rows <= 58
                 else f"compress_rows_fp32_kernel<{bits}, float>")
    kernels.append({"kernel": rows_name + " (V rows: select + fill + quantize + pack + error)",
                    "ms": ms_rows, "alg_bytes": b_rows, "not_counted": "the fp16 error it also writes (2n bytes): an intermediate"})
    del errb, rows_out
    # (2) the fused K path, kernel by kernel (variant hooks of gear_compress_key_fused)
    if C.key_fused_supported(T, D, group, bits, k_key):
        ms_full = timed(lambda: C.compress_key_fused(K, bits, group, k_key, rnk, loop, "fp32", P0k))
        ms_sel = timed(lambda: C.compress_key_fused(K, bits, group, k_key, rnk, loop, "fp32", P0k, variant=32)) if k_key else 0.0
        ms_main = timed(lambda: C.compress_key_fused(K, bits, group, k_key, rnk, loop, "fp32", P0k, variant=8 | 16))
        b_sel = 2 * n + BH * D * 2 * k_key * 6 + n / 8
        b_main = 2 * n + n / 8 * (1 if k_key else 0) + n * bits / 8 + 4 * n / group
        kernels.append({"kernel": "k_select_kernel + k_select_fix_kernel (K: per-channel outlier selection over T, token-major input)",
                        "ms": ms_sel, "alg_bytes": b_sel})
        kernels.append({"kernel": f"k_dense_kernel<{bits}, {group}> (K: fused fill + quantize + pack + Gram on the matrix cores; 128-token slabs through two LDS buffers by LDS-DMA)",
                        "ms": ms_main, "alg_bytes": b_main,
                        "not_counted": "the partial Gram matrices (64 KB per head and workgroup); no error matrix is written: the Q pass rebuilds it"})
        ms_kmain = timed(lambda: C.compress_key_fused(K, bits, group, k_key, rnk, loop, "fp32", P0k, variant=8 | 16 | 128))
        kernels.append({"kernel": f"k_main_kernel<{bits}, 1, {group}, float, ...> (rounds 2 - 5: the same work on register-resident tiles with outlier masks; option kfused_main, NOT the default)",
                        "ms": ms_kmain, "alg_bytes": b_main})
        kernels.append({"kernel": "fused K chain (select, main, per-head solve, Q pass)", "ms": ms_full, "alg_bytes": alg["k_compress"]})
        # the single-read alternative (csrc/kone.hip, option kfused_one; round 6): selection + dense part + Gram in
        # ONE launch with one read of K -- measured beside the chain in every run, not the default (profiles/r6_kone.md says why)
        from gear_amd import _lib as _L
        _lib = _L.load()
        if _lib.gear_set_option(b"kfused_one", 1) == 0:
            try:
                ms_one = timed(lambda: C.compress_key_fused(K, bits, group, k_key, rnk, loop, "fp32", P0k, variant=16))
                ms_chain_to_gram = timed(lambda: (_lib.gear_set_option(b"kfused_one", 0), C.compress_key_fused(K, bits, group, k_key, rnk, loop, "fp32", P0k, variant=16), _lib.gear_set_option(b"kfused_one", 1)))
                kernels.append({"kernel": "k_one_kernel (K: selection + fill + quantize + pack + Gram in ONE launch, one read of K; option kfused_one = 1, NOT the default)",
                                "ms": ms_one, "alg_bytes": b_main + BH * D * 2 * k_key * 6,
                                "chain_ms_up_to_the_gram_matrices": ms_chain_to_gram,
                                "exchange_timeouts": _lib.gear_kone_timeouts(), "fallback_heads": _lib.gear_kone_fallback_heads(),
                                "evidence": "profiles/r6_kone.md (phase clocks, PMC traffic 2.68 GB against 3.09 GB)"})
            finally:
                _lib.gear_set_option(b"kfused_one", 0)
    for kx in kernels:
        kx["achieved"] = kx["alg_bytes"] / (kx["ms"] * 1e-3) / 1e9 if kx["ms"] else None
        kx["frac"] = kx["achieved"] / HBM_PEAK_GBS if kx["ms"] else None
    dom = max(kernels[:3], key=lambda kx: kx["ms"])
    # HBM bytes from PMC counters and in-step kernel durations: only from a profile taken on exactly this library (the newest
    # profiles/r*_traffic.json whose lib_sha256 is the loaded library's; tools/make_traffic.py writes it)
    import glob
    traffic, tnote, prof, dom_step_us = None, "no profile for this library build", {}, None
    if world == 1 and not args.layers and not args.emulate_world:
        sha = lib_sha256()
        for tp in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
            cand = json.load(open(tp))
            if cand.get("lib_sha256") == sha and cand.get("config") == args.config:
                prof, tname = cand, os.path.relpath(tp, ROOT)
                break
        if prof:
            base = dom["kernel"].split(" ")[0].split("<")[0]       # compress_rows_wave_kernel / compress_rows_fp32_kernel / k_select_kernel / k_dense_kernel
            same = lambda kname: kname.split("<")[0] == base or (base == "k_select_kernel" and kname.split("<")[0] == "k_select_fix_kernel")
            hit = [tb for kname, tb in prof.get("kernels", {}).items() if same(kname)]
            # (both instantiations of the wave-per-row kernel -- fast and fallback pass -- share the base name and are added up)
            if hit:
                traffic, tnote = float(sum(hit)), f"{tname} ({prof.get('how', '')})"
            st = [us for kname, us in prof.get("bench_step_avg_us", {}).items() if same(kname)]
            dom_step_us = float(sum(st)) if st else None
        else:
            tnote = "no profiles/r*_traffic.json was measured on this library build / config"
    dominant = {"bound": "hbm", "kernel": dom["kernel"], "achieved": dom["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": dom["frac"], "traffic": traffic, "traffic_source": tnote, "alg_bytes_per_launch": dom["alg_bytes"],
                "ms_per_launch": dom["ms"],
                # the same kernel inside the two-stream bench step (rocprofv3 trace of this command, average per launch, all of
                # its instantiations): the K and V chains share the chip there
                "ms_per_launch_in_step": dom_step_us / 1e3 if dom_step_us else None,
                "frac_in_step": dom["alg_bytes"] / (dom_step_us * 1e-6) / 1e9 / HBM_PEAK_GBS if dom_step_us else None,
                "note": "ms_per_launch: timed alone, back to back on one stream (HIP events); ms_per_launch_in_step: from the "
                        "rocprofv3 kernel trace of the bench step (profiles/*_kernel_stats_bench.md), null without a profile of "
                        "exactly this library"}
    # the headline roofline object is what north_star names -- the fused K / V quant + low-rank + outlier COMPRESS, i.e. the chain
    # of launches per tensor kind -- not its fastest member: the chain with the lower fraction
    cname = min(("k_compress", "v_compress"), key=lambda c: chain[c]["frac"])
    # PMC traffic of the chain = the sum over its launches (same library-tied profile as above)
    chain_traffic = None
    members = {"k_compress": ("k_select_kernel", "k_select_fix_kernel", "k_dense_kernel", "k_solve_kernel", "k_qpass_kernel"),
               "v_compress": ("compress_rows_wave_kernel", "compress_rows_fp32_kernel", "lr_gram_wave_kernel", "k_solve_kernel",
                              "lr_qpass_tm_mfma_kernel")}
    if traffic is not None:
        hit = [tb for kname, tb in prof.get("kernels", {}).items()
               if any(kname == mname or kname.startswith(mname + "<") or kname.startswith(mname) and "<" in mname for mname in members[cname])]
        if hit:
            chain_traffic = float(sum(hit))
    launches = {"k_compress": "k_select_kernel + k_select_fix_kernel + k_dense_kernel + k_solve_kernel + k_qpass_kernel",
                "v_compress": "compress_rows_wave_kernel (fast + fallback pass) + lr_gram_wave_kernel + k_solve_kernel + lr_qpass_tm_mfma_kernel"}
    roofline = {"bound": "hbm", "kernel": f"{cname} chain: {launches[cname]}", "achieved": chain[cname]["achieved"],
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": chain[cname]["frac"], "traffic": chain_traffic,
                "traffic_source": tnote + " (sum over the chain's kernels; per kernel: the *_pmc_traffic.md beside it)",
                "alg_bytes_per_launch": chain[cname]["alg_bytes"],
                "ms_per_launch": chain[cname]["ms"], "launch": "one chain = one call of gear_compress_%s_fused over all layers" % ("key" if cname == "k_compress" else "value"),
                "bytes_definition": "SURVEY.md 8(d) verbatim: read 2n + codes n*b/8 + scale/mn 4n/g + factors 2r(T+D) per head + 6 bytes per outlier; no error term (stored_bytes in roofline_chain = what the build writes: fp32 scale/mn, uint16 indices)",
                "dominant_kernel": dominant,
                "spread": "the same binary measures within +-4 % from box to box (round-1 observation, DESIGN.md section 6)"}

    # ---- decompress-into-attention: one decode token's attention over the compressed cache of ALL layers
    from gear_amd.attention import decode_attention
    from gear_amd import parallel
    n_rep = cfg["q_heads"] // cfg["kv_heads"]
    qv = torch.randn((layers, Hl * n_rep, 1, D), device=dev, dtype=torch.float16)

    def attn_step():
        o = decode_attention(qv, pk, pv)
        if dist is not None:       # head shards -> full [layers, 1, Hq*D] on every rank (latency-bound, a few KiB / layer)
            o = parallel.all_gather_heads(o.transpose(1, 2).reshape(layers, 1, Hl * n_rep * D), world)
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
    payload_ratio = (2 * n * 2) / (pk.nbytes() + pv.nbytes())

    res = None
    if rank == 0:
        res = {
            "metric": "KV compress+decompress GB/s (fp16 KV bytes through compress + through decompress per second)",
            "value": value, "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            # untimed steps IN FRONT of the `warmup` steps of the contract (device clocks / power state; --prewarm-s 0 = none)
            "prewarm_steps": prewarm_steps, "prewarm_s": round(prewarm_seconds, 3),
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32 arithmetic on fp16 data, int%d payload" % bits, "data": "synthetic",
            "config": {"workload": f"{cfg['model']} KV cache, {layers} layers x {H} KV heads x T={T} x D={D}, "
                                   f"{bits}-bit g={group} per-channel K / per-token V + rank-{rnk} (loop {loop}) + "
                                   f"{sparsity * 100:.0f}% outliers (BASELINE configs[{cfg['idx']}])",
                       "parallelism": f"head-shard x{world}", "k_outliers_per_side": [k_key, k_val],
                       "streams": args.streams, "key_path": key_path if key_path != "auto" else
                       ("fused" if C.key_fused_supported(T, D, group, bits, k_key) and C._KEY_PATH_DEFAULT == "auto" else C._KEY_PATH_DEFAULT)},
            "compress_GBps": fp16_bytes_job / ((stages["k_compress"] + stages["v_compress"]) * 1e-3) / 1e9,
            "decompress_GBps": fp16_bytes_job / ((stages["k_decompress"] + stages["v_decompress"]) * 1e-3) / 1e9,
            "stage_ms": stages,
            "timing_note": "value / ms_per_step: wall time of steps whose K and V chains overlap on %d HIP streams; stage_ms, "
                           "compress_GBps, decompress_GBps and roofline_chain: a separate serial pass, one chain at a time "
                           "(their sum exceeds ms_per_step)" % n_str,
            "payload_ratio": payload_ratio,
            "attn_decode": {"ms_per_token_all_layers": attn_ms, "compressed_GBps": payload_bytes_job / (attn_ms * 1e-3) / 1e9,
                            "fp16_equiv_GBps": fp16_bytes_job / (attn_ms * 1e-3) / 1e9,
                            "collective": "all_gather of per-rank attention output" if world > 1 else None},
            "roofline": roofline,
            "roofline_chain": chain,
            "kernels": kernels,
        }
        if shard_info is not None:
            if dt_other is not None:
                shard_info["per_shard"]["value_GBps"] = 2 * fp16_bytes_job * args.steps / dt_other / 1e9
                shard_info["per_shard"]["ms_per_step"] = dt_other / args.steps * 1e3
                res["value_per_shard_selection"] = shard_info["per_shard"]["value_GBps"]
            res["sharding"] = shard_info
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(cfg)
    del K, V, kr, vr, pk, pv, out, qv
    torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.layers:
        res["block_boundary"] = block_boundary_cost(cfg, dev, Hl)
        res["block_boundary"]["traffic"] = None
        if traffic is not None and not args.emulate_world:
            bt = [tb for kname, tb in prof.get("kernels", {}).items() if kname.startswith("block_compress_kernel")]
            res["block_boundary"]["traffic"] = float(sum(bt)) if bt else None
    if rank == 0 and world == 1 and not args.layers and not args.emulate_world:
        res["attn_decode"]["one_layer_streaming_cache_by_batch"] = attn_decode_by_batch(cfg, dev)
    if not args.no_decode and not args.layers and not args.emulate_world:
        try:
            dec = decode_tokens_per_s(cfg, dev, world, rank, args.decode_tokens, args.exchange)     # (every rank takes part when sharded)
        except Exception as e:           # the decode leg is a secondary figure: the line with `value` must still come out
            oom = isinstance(e, torch.cuda.OutOfMemoryError)
            if world == 1 and not oom:
                raise
            # (config 5 on ONE GPU: 70B weights + the decoder's fused copies exceed 288 GB -- the leg is skipped, not the line)
            dec = {"error": f"rank {rank}: {type(e).__name__}: {str(e)[:300]}", "tokens_per_s": 1e-9}
            if oom:
                dec["skipped"] = "out of memory: the model's weights and the decoder's fused copies do not fit this GPU"
                torch.cuda.empty_cache()
        if dist is not None:
            t = torch.tensor([1.0 / dec["tokens_per_s"]], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)                           # slowest rank
            dec["tokens_per_s"] = 1.0 / float(t.item())
            dec["ms_per_token"] = 1e3 * float(t.item())
        if rank == 0:
            if "block_boundary" in res:
                dec["block_compress_ms"] = res["block_boundary"]["block_kernel_us"] * 1e-3
            res["decode"] = dec
    if rank == 0:
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
