#!/bin/bash
# The whole evidence set of a round from ONE box and ONE library build (run on the GPU box from the repo root):
# tools/collect_profiles.sh for config 3 and config 2, the traffic tables published from their PMC passes, then bench.py AGAIN for
# both configs so that the committed bench lines carry `roofline.traffic` of this very library, the emulated shard lines, and the
# single-GPU lines of configs 4 and 5.  Everything lands in gpurun_out/<tag>_published/ (copy its contents into profiles/).
# usage: tools/collect_all.sh <tag>        e.g. tools/collect_all.sh r5
set -u
TAG=$1
PUB=gpurun_out/${TAG}_published
mkdir -p $PUB
tools/collect_profiles.sh $TAG > gpurun_out/collect_$TAG.log 2>&1
GEAR_PROF_CONFIG=c2 tools/collect_profiles.sh ${TAG}c2 --config c2 > gpurun_out/collect_${TAG}c2.log 2>&1
python tools/publish_profiles.py $TAG c3
python tools/publish_profiles.py ${TAG}c2 c2
python bench.py 2> $PUB/bench.err | tail -1 > profiles/${TAG}_bench_line.json
python bench.py --config c2 2>> $PUB/bench.err | tail -1 > profiles/${TAG}_bench_c2_line.json
rm -f profiles/${TAG}c2_bench_line.json
python bench.py --config c4 --no-cpu-baseline 2>> $PUB/bench.err | tail -1 > profiles/${TAG}_bench_c4_1gpu_line.json
python bench.py --config c5 --no-cpu-baseline --no-decode 2>> $PUB/bench.err | tail -1 > profiles/${TAG}_bench_c5_1gpu_line.json
tools/emu_lines.sh profiles/${TAG}_emulation.jsonl > $PUB/emu.log 2>&1
cp profiles/${TAG}_* profiles/${TAG}c2_* $PUB/
ls -la $PUB | tail -40
