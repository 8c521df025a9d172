#!/bin/bash
# SQ instruction counters of the row compressor for each experiment flag (GPU box, from the repo root)
export TMPDIR=/tmp
OUT=gpurun_out/rows_ab; mkdir -p $OUT
for f in "$@"; do
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/f$f -o pmc -- python tools/exp_rows_ab.py $f > $OUT/f$f.log 2>&1
  echo "== flag $f"; python tools/pmc_table.py $(find $OUT/f$f -name "*counter_collection.csv" | head -1) | grep -A9 "compress_rows" | cut -c1-100
  rm -rf $OUT/f$f
done
