"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel name (short) and counter, the per-dispatch values."""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
tab = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\(.*", "", n)[:60]
    tab.setdefault(n, collections.OrderedDict()).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for n, cs in tab.items():
    if not n.startswith(("compress_rows", "lr_", "decompress", "attn", "k_select", "k_main", "k_dense", "k_solve", "k_qpass", "transpose_f16")):
        continue
    print("##", n)
    for c, v in cs.items():
        print(f"  {c:24s}", " ".join(f"{x:.3e}" for x in v[:9]))
