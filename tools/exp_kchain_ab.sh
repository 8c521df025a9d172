#!/bin/bash
# A/B of library builds on the fused K chain (tools/exp_kchain.py), GPU box: tools/_ab/<name>.so swapped in.
cd "$(dirname "$0")/.."
cp gear_amd/libgear_hip.so /tmp/cur.so
for rep in 1 2; do
for v in "$@"; do
  if [ $v = cur ]; then cp /tmp/cur.so gear_amd/libgear_hip.so; else cp tools/_ab/$v.so gear_amd/libgear_hip.so; fi
  echo -n "== $v: "; python tools/exp_kchain.py 2>&1 | tail -1
done
done
cp /tmp/cur.so gear_amd/libgear_hip.so
