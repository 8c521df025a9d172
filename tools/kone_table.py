"""Markdown table of tools/pmc_kone.sh's passes: per kernel of the K compress chain the average duration (kernel trace) and the HBM
bytes per launch (FETCH_SIZE x 2 / WRITE_SIZE, KB units, calibrated on the 1 GiB copy of the same run)."""
import csv, re, sys, collections
out = sys.argv[1]


def table(path):
    tab = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        n = re.sub(r"\(.*", "", n)
        tab.setdefault(n, []).append(float(r["Counter_Value"]))
    return tab


def med(v):
    return sorted(v)[len(v) // 2]


fetch, write = table(f"{out}/pmc_FETCH_SIZE.csv"), table(f"{out}/pmc_WRITE_SIZE.csv")
GiB = float(1 << 30)
cal = [n for n in fetch if "copy" in n.lower() or "elementwise" in n.lower()]
kf = GiB / max(max(fetch[n]) for n in cal)
kw = GiB / max(max(write[n]) for n in cal)
# durations from the kernel trace: the chain's kernels are also launched behind the single-read kernel, where they return at once
# (no head needed the exact fall-back) -- the three LONGEST dispatches of a kernel are its real launches
import glob
durs = collections.OrderedDict()
for r in csv.DictReader(open(glob.glob(f"{out}/trace/*kernel_trace.csv")[0])):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\(.*", "", n)
    durs.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
dur = {n: (len(v), sum(sorted(v)[-3:]) / min(3, len(v)), min(v)) for n, v in durs.items()}
print(f"calibration: {kf:.1f} B per FETCH_SIZE unit, {kw:.1f} B per WRITE_SIZE unit (1 GiB copy)\n")
print("| kernel | launches | us (3 longest) | us (shortest) | read GB | written GB | total GB |\n|---|---:|---:|---:|---:|---:|---:|")
for n in fetch:
    if not n.startswith(("k_select", "k_main", "k_dense", "k_solve", "k_qpass", "k_one")):
        continue
    f, w = max(fetch[n]) * kf, max(write.get(n, [0.0])) * kw
    c, us, us_min = dur.get(n, (0, 0.0, 0.0))
    print(f"| `{n[:70]}` | {c} | {us:.1f} | {us_min:.1f} | {f / 1e9:.3f} | {w / 1e9:.3f} | {(f + w) / 1e9:.3f} |")
