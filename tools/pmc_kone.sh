#!/bin/bash
# Kernel trace + HBM traffic (FETCH_SIZE, WRITE_SIZE in separate PMC passes) of the K compress chain, kernel chain against the
# single-read kernel (tools/prof_kone.py).  Run on the GPU box from the repo root: tools/pmc_kone.sh gpurun_out/kone
set -u
OUT=$1
export TMPDIR=/tmp
mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python tools/prof_kone.py > $OUT/trace.log 2>&1
for P in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/$P -o pmc -- python tools/prof_kone.py > $OUT/$P.log 2>&1
  cp $(find $OUT/$P -name "*counter_collection.csv" | head -1) $OUT/pmc_$P.csv
done
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python tools/kone_table.py $OUT
