#!/bin/bash
# A/B of bench.py on ONE box: each argument is an environment assignment list ("GEAR_KFUSED_MAIN=1", "" = defaults); every
# configuration runs $REPS times, interleaved.  usage: tools/bench_ab.sh "" "GEAR_KFUSED_MAIN=1"
REPS=${REPS:-2}
mkdir -p gpurun_out
for r in $(seq $REPS); do
  i=0
  for cfg in "$@"; do
    env $cfg timeout 600 python bench.py ${BENCH_ARGS:-} > gpurun_out/ab_${i}_${r}.json 2> gpurun_out/ab_${i}_${r}.err
    python - "$cfg" gpurun_out/ab_${i}_${r}.json <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print("[%s] value %.1f ms/step %.4f stages %s" % (sys.argv[1], d["value"], d["ms_per_step"], {k: round(v, 4) for k, v in d["stage_ms"].items()}), flush=True)
except Exception as e:
    print("[%s] failed: %r" % (sys.argv[1], e), flush=True)
P
    i=$((i+1))
  done
done
