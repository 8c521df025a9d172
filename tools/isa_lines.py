"""Static VALU / SALU / memory instruction counts of one kernel per source line.
usage: hipcc <flags> -gline-tables-only -S --cuda-device-only file.hip -o out.s ; python tools/isa_lines.py out.s <mangled-name-substring> [file.hip]
Counts are static (a loop body counts once); use them to see where the instruction budget of the straight-line parts goes."""
import re, sys, collections

asm, key = sys.argv[1], sys.argv[2]
src = open(sys.argv[3]).read().split("\n") if len(sys.argv) > 3 else None
lines = open(asm).read().split("\n")
start = next(i for i, l in enumerate(lines) if key in l and re.match(r"^_Z\S+:", l))
end = next(i for i in range(start, len(lines)) if ".Lfunc_end" in lines[i])
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = m.group(3) or m.group(2)
cur = None
cnt = collections.defaultdict(lambda: [0, 0, 0])
for l in lines[start:end]:
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (int(m.group(1)), int(m.group(2)))
        continue
    t = l.strip()
    if t.startswith("v_"):
        cnt[cur][0] += 1
    elif t.startswith("s_"):
        cnt[cur][1] += 1
    elif t.startswith(("ds_", "global_", "buffer_", "flat_", "scratch_")):
        cnt[cur][2] += 1
tot = [sum(c[i] for c in cnt.values()) for i in range(3)]
print(f"total VALU {tot[0]} SALU {tot[1]} mem {tot[2]}")
for (f, ln), c in sorted(cnt.items(), key=lambda kv: (kv[0] is None, kv[0])):
    name = files.get(f, str(f)).split("/")[-1]
    text = ""
    if src and name == sys.argv[3].split("/")[-1] and 0 < ln <= len(src):
        text = src[ln - 1].strip()[:90]
    print(f"{name}:{ln:5d}  v={c[0]:4d} s={c[1]:4d} m={c[2]:3d}  {text}")
