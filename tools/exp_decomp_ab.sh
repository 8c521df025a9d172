#!/bin/bash
# A/B of decompress_rows builds on one box: tools/_ab/*.so are swapped in as the library (same sources hash file), GPU box only.
# usage: tools/exp_decomp_ab.sh <variant[:ENV=VAL]> ...      (variant "cur" = the library in the tree)
cd "$(dirname "$0")/.."
cp gear_amd/libgear_hip.so /tmp/cur.so
for rep in 1 2; do
for spec in "$@"; do
  v=${spec%%:*}; e=""; [ "$spec" != "$v" ] && e=${spec#*:}
  if [ $v = cur ]; then cp /tmp/cur.so gear_amd/libgear_hip.so; else cp tools/_ab/$v.so gear_amd/libgear_hip.so; fi
  echo "== $spec"
  env $e python tools/exp_decomp.py 2>&1 | grep -E "decompress" | tr '\n' ';' | sed 's/decompress //g; s/ ms//g'; echo
done
done
cp /tmp/cur.so gear_amd/libgear_hip.so
