#!/bin/bash
# One rank's shard of the multi-GPU configurations, EMULATED on one GPU (no collective is on the compress / decompress path):
# bench lines for profiles/<tag>_emulation.jsonl.  These are not scaling measurements.  usage: tools/emu_lines.sh <out.jsonl>
OUT=$1; : > $OUT
for spec in "c3 1" "c3 2" "c3 4" "c3 8" "c4 4" "c5 8"; do
  set -- $spec
  python bench.py --no-cpu-baseline --no-decode --config $1 --emulate-world $2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
keep={k:d[k] for k in ('metric','value','unit','ms_per_step','stage_ms','config','roofline_chain')}
keep['block_boundary']=d.get('block_boundary')
keep['sharding']=d.get('sharding')
keep['value_per_shard_selection']=d.get('value_per_shard_selection')
keep['emulation']='one rank of a %s-way head shard of config %s, run alone on one GPU: NOT a multi-GPU measurement' % (sys.argv[2], sys.argv[1])
print(json.dumps(keep))" $1 $2 >> $OUT
done
cat $OUT | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); sh=d.get('sharding') or {}; print(d['emulation'][:44], round(d['ms_per_step'],3), 'per-shard selection:', (sh.get('per_shard') or {}).get('ms_per_step'), 'parity', sh.get('shard_parity'), {k:round(v,3) for k,v in d['stage_ms'].items()}, (d.get('block_boundary') or {}).get('block_kernel_us'))"
