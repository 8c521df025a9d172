"""Print the top kernels of a rocprofv3 --kernel-trace --stats run as a markdown table (short names)."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 20
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z_0-9:]+(<[^()]*?>)?)\(", n)
    n = m.group(1) if m else n
    return n[:90]
print("| kernel | calls | avg us | total ms | % |\n|---|---:|---:|---:|---:|")
for r in rows[:top]:
    print(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['Percentage']):.1f} |")
