#!/bin/bash
# per-rank step time of an N-way head shard, emulated on one GPU (debug aid for the strong-scaling path)
for n in 1 2 4 8; do
  python bench.py --no-cpu-baseline --no-decode --emulate-world $n 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['stage_ms'].items()})" $n
done
