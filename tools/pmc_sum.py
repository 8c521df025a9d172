"""Print per-dispatch values of one counter from a rocprofv3 --pmc counter_collection.csv, grouped by short kernel name."""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
tab = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\(.*", "", n)[:70]
    tab.setdefault((n, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
for (n, c), v in tab.items():
    print(f"{n:70s} {c:12s}", " ".join(f"{x:.4e}" for x in v))
