#!/bin/bash
# The PMC traffic passes of tools/collect_profiles.sh alone (FETCH_SIZE, WRITE_SIZE over tools/prof_step.py) + the library hash:
# refreshes gpurun_out/<tag>/pmc_p3.csv / pmc_p4.csv / lib.sha256 after a kernel change that leaves the other tables valid.
set -u
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
i=2
for P in FETCH_SIZE WRITE_SIZE; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$i -o pmc -- python tools/prof_step.py > $OUT/p$i.log 2>&1
  cp $(find $OUT/p$i -name "*counter_collection.csv" | head -1) $OUT/pmc_p$i.csv
  rm -rf $OUT/p$i
done
sha256sum gear_amd/libgear_hip.so > $OUT/lib.sha256
