#!/bin/bash
# Round evidence on the GPU box (run from the repo root): bench line, rocprofv3 kernel stats of the same command, PMC passes
# (SQ counters; FETCH_SIZE and WRITE_SIZE in their own passes) over the compress kernels at full size.  Writes gpurun_out/<tag>/.
# usage: tools/collect_profiles.sh <tag> [bench args...]
set -u
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py "$@" > $OUT/bench_line.json 2> $OUT/bench.err
tail -c 600 $OUT/bench_line.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o st -- python bench.py --no-cpu-baseline --no-decode "$@" > $OUT/stats.log 2>&1
python tools/kstats.py $(find $OUT/stats -name "*kernel_stats.csv" | head -1) 30 > $OUT/kernel_stats_bench.md
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_dec -o st -- python tools/prof_decode.py > $OUT/stats_dec.log 2>&1
python tools/kstats.py $(find $OUT/stats_dec -name "*kernel_stats.csv" | head -1) 30 > $OUT/kernel_stats_decode.md
# (the script's own result lines only: rocprofv3 writes its log to the same stream)
grep -v -E "^[EWI][0-9]{8} " $OUT/stats_dec.log | grep -E "^(ms/token|enqueue)" | sed "s/^/\n/" | tail -4 >> $OUT/kernel_stats_decode.md
# the compress kernels ONE AT A TIME on one stream (tools/prof_step.py): the durations the per-kernel roofline fractions stand on
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_iso -o st -- python tools/prof_step.py > $OUT/stats_iso.log 2>&1
python tools/kstats.py $(find $OUT/stats_iso -name "*kernel_stats.csv" | head -1) 30 > $OUT/kernel_stats_isolated.md
# the decode-time block boundary: single-launch block compressor vs the chain (tools/prof_block.py)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_blk -o st -- python tools/prof_block.py > $OUT/stats_blk.log 2>&1
python tools/kstats.py $(find $OUT/stats_blk -name "*kernel_stats.csv" | head -1) 30 > $OUT/kernel_stats_block.md
for P in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pb_$P -o pmc -- python tools/prof_block.py > $OUT/pb_$P.log 2>&1
  cp $(find $OUT/pb_$P -name "*counter_collection.csv" | head -1) $OUT/pmc_block_$P.csv
  rm -rf $OUT/pb_$P
done
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
P2="SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"
i=0
for P in "$P1" "$P2" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$i -o pmc -- python tools/prof_step.py > $OUT/p$i.log 2>&1
  cp $(find $OUT/p$i -name "*counter_collection.csv" | head -1) $OUT/pmc_p$i.csv
  rm -rf $OUT/p$i
done
rm -rf $OUT/stats $OUT/stats_dec $OUT/stats_iso $OUT/stats_blk
sha256sum gear_amd/libgear_hip.so > $OUT/lib.sha256
ls -la $OUT
