"""Turn the PMC passes of tools/collect_profiles.sh into profiles/<tag>_traffic.json + a markdown table.
HBM bytes per launch = FETCH_SIZE x 2 x 1024 / ... : on gfx950 rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB and FETCH_SIZE
counts half of what is read (MI355X_MICROARCH.md, HBM / rocprofv3 section), calibrated here on a 1 GiB copy in the same run.
usage: python tools/make_traffic.py gpurun_out/<tag> profiles/<tag> <config>"""
import csv, json, re, sys, collections

src, dst, config = sys.argv[1], sys.argv[2], sys.argv[3]


def table(path):
    tab = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        n = re.sub(r"\(.*", "", n)
        tab.setdefault(n, collections.OrderedDict()).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return tab


fetch, write = table(f"{src}/pmc_p3.csv"), table(f"{src}/pmc_p4.csv")
GiB = float(1 << 30) * (0.5 if config == "c2" else 1.0)     # (the calibration copy is the tensor itself: 1 GiB at config 3, 0.5 at config 2)
# calibration: the copy kernel of the 1 GiB y.copy_(x)
cal = [n for n in fetch if "copy" in n.lower() or "elementwise" in n.lower()]
cal_f = max(max(fetch[n]["FETCH_SIZE"]) for n in cal)
cal_w = max(max(write[n]["WRITE_SIZE"]) for n in cal)
kf, kw = GiB / cal_f, GiB / cal_w       # bytes per counter unit, from the copy
out = {"lib_sha256": open(f"{src}/lib.sha256").read().split()[0], "config": config, "kernels": {},
       "how": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over tools/prof_step.py; bytes per counter unit calibrated on a "
              f"1 GiB copy in the same run (read {kf:.1f}, write {kw:.1f} B / unit: KB units, FETCH_SIZE counts half the bytes)"}
rows = []
for n in fetch:
    if not n.startswith(("compress_rows", "k_select", "k_main", "k_dense", "k_solve", "k_qpass", "lr_", "transpose_f16", "decompress_rows")):
        continue
    f = sorted(fetch[n]["FETCH_SIZE"])[len(fetch[n]["FETCH_SIZE"]) // 2] * kf
    w = sorted(write[n]["WRITE_SIZE"])[len(write[n]["WRITE_SIZE"]) // 2] * kw
    rows.append((n, f, w))
    short = re.sub(r"<.*", "", n)
    out["kernels"][n[:60]] = f + w
# average duration of the same kernels INSIDE the two-stream bench step (kernel_stats_bench.md of the same collection: the rocprofv3
# trace of `python bench.py`, whose launches are overwhelmingly the step's) -- bench.py's roofline.dominant_kernel.ms_per_launch_in_step
import os
ksb = f"{src}/kernel_stats_bench.md"
if os.path.exists(ksb):
    out["bench_step_avg_us"] = {}
    for ln in open(ksb):
        m = re.match(r"\| `([^`]+)` \| (\d+) \| ([0-9.]+) \|", ln)
        if m and m.group(1).startswith(("compress_rows", "k_select", "k_main", "k_dense", "k_solve", "k_qpass", "lr_", "decompress_rows")):
            out["bench_step_avg_us"][m.group(1)[:60]] = float(m.group(3))
json.dump(out, open(f"{dst}_traffic.json", "w"), indent=1)
with open(f"{dst}_pmc_traffic.md", "w") as fmd:
    fmd.write(f"# HBM traffic per launch (PMC), config {config}, library {out['lib_sha256'][:12]}\n\n{out['how']}\n\n"
              "| kernel | read GB | written GB | total GB |\n|---|---:|---:|---:|\n")
    for n, f, w in rows:
        fmd.write(f"| `{n[:80]}` | {f / 1e9:.3f} | {w / 1e9:.3f} | {(f + w) / 1e9:.3f} |\n")
# the block boundary (tools/prof_block.py), when collected: same calibration
import os
if os.path.exists(f"{src}/pmc_block_FETCH_SIZE.csv"):
    bf, bw_ = table(f"{src}/pmc_block_FETCH_SIZE.csv"), table(f"{src}/pmc_block_WRITE_SIZE.csv")
    with open(f"{dst}_pmc_traffic.md", "a") as fmd:
        fmd.write("\n## The decode-time block boundary (tools/prof_block.py: 32 layers x 32 heads x 64 tokens, K and V = 33.6 MB of fp16 in)\n\n"
                  "| kernel | read MB | written MB | total MB |\n|---|---:|---:|---:|\n")
        for n in bf:
            if not n.startswith(("block_compress", "compress_rows", "k_select", "k_main", "k_dense", "k_solve", "k_qpass", "lr_", "ktile", "vtile", "outlier_chunk")):
                continue
            f = sorted(bf[n]["FETCH_SIZE"])[len(bf[n]["FETCH_SIZE"]) // 2] * kf
            w = sorted(bw_[n]["WRITE_SIZE"])[len(bw_[n]["WRITE_SIZE"]) // 2] * kw if n in bw_ else 0.0
            fmd.write(f"| `{n[:80]}` | {f / 1e6:.2f} | {w / 1e6:.2f} | {(f + w) / 1e6:.2f} |\n")
            if n.startswith("block_compress"):
                out["kernels"][n[:60]] = f + w
    json.dump(out, open(f"{dst}_traffic.json", "w"), indent=1)
print(open(f"{dst}_pmc_traffic.md").read())
