"""Copy the evidence of tools/collect_profiles.sh from gpurun_out/<tag>/ into profiles/ (tracked): the bench line, the rocprofv3
kernel-stats tables, the SQ counter tables of the compress kernels and the PMC traffic table + profiles/<tag>_traffic.json.
usage: python tools/publish_profiles.py <tag> <config>"""
import json, os, subprocess, sys

tag, config = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles", tag)
sha = open(os.path.join(src, "lib.sha256")).read().split()[0]
line = open(os.path.join(src, "bench_line.json")).read().strip().splitlines()[-1]
json.loads(line)
open(dst + "_bench_line.json", "w").write(line + "\n")
hdr = {"bench": "`rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-decode` (default N=1, config c3; the "
                "run also contains the warm-up steps, the per-kernel timing passes and the attention-by-batch calls)",
       "decode": "`rocprofv3 --kernel-trace --stats -- python tools/prof_decode.py` (FastGearDecoder, Llama-2-7B shapes, prompt 4040, "
                 "2 % outliers, 20 + 20 timed eager token steps after the prefill; hipBLASLt kernels = the prefill GEMMs)"}
hdr["isolated"] = ("`rocprofv3 --kernel-trace --stats -- python tools/prof_step.py`: the compress kernels of one bench step at config-3 size, "
                   "ONE AT A TIME on one stream (3 launches each after a 1 GiB copy) -- the durations bench.py's `kernels` fractions stand "
                   "on; inside the bench step the K and V chains share the chip on two streams and the same kernels take longer")
hdr["block"] = ("`rocprofv3 --kernel-trace --stats -- python tools/prof_block.py`: the decode-time block boundary at Llama-2-7B shapes "
                "(32 layers x 32 heads x 64 tokens, K and V, 2 % outliers, rank 8): `block_compress_kernel` (ONE launch per boundary) x 5, "
                "then the kernel chain it replaced x 5 (one launch of each kernel below per boundary)")
for kind in ("bench", "decode", "isolated", "block"):
    if not os.path.exists(os.path.join(src, f"kernel_stats_{kind}.md")):
        continue
    body = open(os.path.join(src, f"kernel_stats_{kind}.md")).read()
    open(dst + f"_kernel_stats_{kind}.md", "w").write(
        f"# rocprofv3 kernel stats, {tag}, library {sha[:12]}\n\n{hdr[kind]}\n\n{body}")
with open(dst + "_pmc_sq_compress.md", "w") as f:
    f.write(f"# rocprofv3 --pmc SQ counters over the compress kernels at config-3 size, {tag}, library {sha[:12]}\n\n"
            "`rocprofv3 --kernel-trace --pmc <8 counters> -- python tools/prof_step.py` (two passes; 3 launches per kernel, per-dispatch values).\n"
            "VALU instructions per wave = SQ_INSTS_VALU / SQ_WAVES; VALU busy = SQ_ACTIVE_INST_VALU x 4 cycles / (1024 SIMDs x duration x 2.4 GHz).\n\n```\n")
    for p in ("pmc_p1.csv", "pmc_p2.csv"):
        f.write(subprocess.run([sys.executable, os.path.join(root, "tools", "pmc_table.py"), os.path.join(src, p)],
                               capture_output=True, text=True).stdout)
    f.write("```\n")
subprocess.check_call([sys.executable, os.path.join(root, "tools", "make_traffic.py"), src, dst, config], stdout=subprocess.DEVNULL)
print("published", tag, sha[:12])
