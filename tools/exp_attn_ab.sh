#!/bin/bash
# A/B of library builds on the decode attention over the streaming cache (tools/exp_attn_stream.py), GPU box.
cd "$(dirname "$0")/.."
cp gear_amd/libgear_hip.so /tmp/cur.so
for rep in 1 2; do
for v in "$@"; do
  if [ $v = cur ]; then cp /tmp/cur.so gear_amd/libgear_hip.so; else cp tools/_ab/$v.so gear_amd/libgear_hip.so; fi
  echo "== $v"; python tools/exp_attn_stream.py 2>&1 | tail -6
done
done
cp /tmp/cur.so gear_amd/libgear_hip.so
