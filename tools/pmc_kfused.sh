#!/bin/bash
# PMC passes (SQ counters, then HBM traffic) over the fused K chain at reduced layer count; run on the GPU box from the repo root.
# usage: tools/pmc_kfused.sh <outdir> [layers]
set -u
OUT=$1; L=${2:-8}
export TMPDIR=/tmp ONLY="fused k+r"
mkdir -p $OUT
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
P2="SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"
P3="FETCH_SIZE"
P4="WRITE_SIZE"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$i -o pmc -- python tools/exp_kfused.py $L > $OUT/p$i.log 2>&1
  python tools/pmc_table.py $(find $OUT/p$i -name "*counter_collection.csv" | head -1) > $OUT/pmc_p$i.txt 2>&1
done
cat $OUT/pmc_p*.txt
