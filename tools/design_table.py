"""Print the "current numbers" table of DESIGN.md section 6 from the committed bench lines (profiles/<tag>_bench_line.json,
<tag>_bench_c2_line.json).  usage: python tools/design_table.py r5"""
import json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r5"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ld = lambda n: json.load(open(os.path.join(root, "profiles", n)))
c3, c2 = ld(f"{tag}_bench_line.json"), ld(f"{tag}_bench_c2_line.json")


def g(d):
    r, de = d["roofline"], d["decode"]
    return d["roofline_chain"], r, r["dominant_kernel"], de, d["attn_decode"]["one_layer_streaming_cache_by_batch"], d["block_boundary"], de["vs_fp16_cache"]


ch, r, dk, de, ad, bb, vs = g(c3)
ch2, r2, dk2, de2, ad2, bb2, vs2 = g(c2)
f = lambda x, n=3: f"{x:.{n}f}"
at = lambda a, k: " / ".join(f"{a[b][k]:.1f}" for b in ("B1", "B4", "B16"))
sp = lambda a: " / ".join(f"{a[b]['speedup_vs_fp16_cache']:.2f}" for b in ("B1", "B4", "B16"))
print(f'''| | config 3 (2-bit, r 8, 2 %, T 4096) | config 2 (4-bit, r 4, 1 %, T 2048; `{tag}_bench_c2_line.json`) |
|---|---:|---:|
| `value` | **{c3["value"]:.0f} GB/s**, {c3["ms_per_step"]:.2f} ms per step | {c2["value"]:.0f} GB/s, {c2["ms_per_step"]:.2f} ms |
| K compress chain (select + fix, main, solve, Q pass) | {f(ch["k_compress"]["ms"])} ms = **{f(ch["k_compress"]["frac"])}**; PMC {r["traffic"]/1e9:.2f} GB = {r["traffic"]/r["alg_bytes_per_launch"]:.1f} x algorithmic | {f(ch2["k_compress"]["ms"])} ms = {f(ch2["k_compress"]["frac"])}; PMC {r2["traffic"]/1e9:.2f} GB |
| V compress chain (rows, Gram, solve, Q pass) | {f(ch["v_compress"]["ms"])} ms = **{f(ch["v_compress"]["frac"])}** | {f(ch2["v_compress"]["ms"])} ms = {f(ch2["v_compress"]["frac"])} |
| K / V decompress | {f(ch["k_decompress"]["ms"])} / {f(ch["v_decompress"]["ms"])} ms = {f(ch["k_decompress"]["frac"])} / {f(ch["v_decompress"]["frac"])} | {f(ch2["k_decompress"]["ms"])} / {f(ch2["v_decompress"]["ms"])} ms = {f(ch2["k_decompress"]["frac"])} / {f(ch2["v_decompress"]["frac"])} |
| dominant kernel (V rows), alone / inside the two-stream step | {f(dk["ms_per_launch"])} ms = {f(dk["frac"])} / {dk["ms_per_launch_in_step"]:.2f} ms = {dk["frac_in_step"]:.3f} (PMC {dk["traffic"]/1e9:.2f} GB = {dk["traffic"]/dk["alg_bytes_per_launch"]:.2f} x) | {f(dk2["ms_per_launch"])} ms = {f(dk2["frac"])} |
| block boundary (one launch) | {bb["block_kernel_us"]:.0f} us (chain {bb["chain_us"]:.0f}) | {bb2["block_kernel_us"]:.0f} us |
| decode, B = 1, 4k context | {de["eager_tokens_per_s"]:.1f} tok/s eager, {de["graph_replay_tokens_per_s"]:.1f} graph-replayed; hook module {de["hook_module_tokens_per_s"]:.0f} | {de2["eager_tokens_per_s"]:.1f} eager, {de2["graph_replay_tokens_per_s"]:.1f} graph-replayed; hook {de2["hook_module_tokens_per_s"]:.0f} |
| same decoder over an fp16 cache, batch 1 / 16 | {vs["B1"]["gear_tokens_per_s"]:.0f} vs {vs["B1"]["fp16_cache_tokens_per_s"]:.0f} ({vs["B1"]["gear_vs_fp16_cache"]:.2f} x) / {vs["B16"]["gear_tokens_per_s"]:.0f} vs {vs["B16"]["fp16_cache_tokens_per_s"]:.0f} tok/s ({vs["B16"]["gear_vs_fp16_cache"]:.2f} x); allocated cache {vs["B1"]["gear_cache_MiB"]/1024:.1f} vs {vs["B1"]["fp16_cache_MiB"]/1024:.1f} / {vs["B16"]["gear_cache_MiB"]/1024:.1f} vs {vs["B16"]["fp16_cache_MiB"]/1024:.1f} GiB | {vs2["B1"]["gear_tokens_per_s"]:.0f} vs {vs2["B1"]["fp16_cache_tokens_per_s"]:.0f} ({vs2["B1"]["gear_vs_fp16_cache"]:.2f} x) / {vs2["B16"]["gear_tokens_per_s"]:.0f} vs {vs2["B16"]["fp16_cache_tokens_per_s"]:.0f} ({vs2["B16"]["gear_vs_fp16_cache"]:.2f} x) |
| one layer's attention, B = 1 / 4 / 16, caches rotated (warm) | {at(ad, "us_per_call")} us ({at(ad, "us_per_call_warm")}) | {at(ad2, "us_per_call")} us ({at(ad2, "us_per_call_warm")}) |
| ... over an fp16 cache of the same length | {at(ad, "fp16_cache_us_per_call")} us -> compression {sp(ad)} x | {at(ad2, "fp16_cache_us_per_call")} us -> {sp(ad2)} x |
| `cpu_baseline` (GPU box host, {c3["cpu_baseline"]["cores"]} threads) | {c3["cpu_baseline"]["value"]:.2f} GB/s | {c2["cpu_baseline"]["value"]:.2f} GB/s |''')
