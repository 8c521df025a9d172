#!/bin/bash
# Multi-GPU preflight: the first time a node with more than one MI355X appears, run this from the repo root.
# For every N in {1, 2, 4, 8} the node has, and for the sharded configurations c3 (7B), c4 (13B, <= 4 ranks: 40 heads), c5 (70B GQA:
# 8 KV heads), it runs `bench.py --gpus N --exchange both` exactly as the driver does (one rank per GPU over RCCL), then checks each
# line: shard_parity true (shard payloads == unsharded payload), the exchange that ran (RCCL all-gather / peer stores), whether the
# RCCL all-gather was CAPTURED in the token-step graph (a fall-back to eager steps fails the preflight: it means the capture
# probe of parallel.HeadGather said no on this node), tokens/s of both exchanges, and the kone / block-kernel status words.
# No number of this script has ever been produced: no multi-GPU node was available to this build in six rounds.
# usage: tools/scale_preflight.sh [outdir]      (exit code 0 = every check passed)
set -u
OUT=${1:-gpurun_out/scale_preflight}
mkdir -p $OUT
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
echo "GPUs visible: $NGPU" | tee $OUT/summary.txt
export HSA_ENABLE_IPC_MODE_LEGACY=0
FAIL=0
PORT=29610
for CFG in c3 c4 c5; do
  for N in 1 2 4 8; do
    [ $N -gt $NGPU ] && continue
    [ $CFG = c4 ] && [ $N -gt 4 ] && continue
    LOG=$OUT/bench_${CFG}_n$N
    EXTRA=""
    [ $CFG = c5 ] && EXTRA="--no-decode"      # (70B weights do not fit beside the fused copies on one GPU; the attention legs run)
    if [ $N -eq 1 ]; then
      python bench.py --gpus 1 --config $CFG --no-cpu-baseline $EXTRA > $LOG.json 2> $LOG.err
    else
      PORT=$((PORT+1))
      python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT \
        bench.py --gpus $N --config $CFG --exchange both --no-cpu-baseline $EXTRA > $LOG.json 2> $LOG.err
    fi
    python - $LOG.json $CFG $N <<'PY' | tee -a $OUT/summary.txt
import json, sys
path, cfg, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
try:
    d = json.loads(open(path).read().strip().splitlines()[-1])
except Exception as e:
    print(f"FAIL {cfg} N={n}: no bench line ({e})"); sys.exit(0)
bad = []
sh = d.get("sharding", {})
if n > 1:
    if sh.get("shard_parity") is not True:
        bad.append("shard_parity is not true")
    dec = d.get("decode", {})
    modes = dec.get("exchange_modes", {})
    col = modes.get("collective", {})
    if dec and not col:
        bad.append("no collective (RCCL) decode leg")
    if col and "graph_replay_tokens_per_s" not in col:
        bad.append("RCCL all-gather NOT captured in the token-step graph (eager steps only): " + str(col.get("exchange")))
    if col and col.get("exchange_class") != "HeadGather":
        bad.append("the collective leg did not run parallel.HeadGather: " + str(col.get("exchange_class")))
    peer = modes.get("peer", {})
    if dec and peer.get("exchange_class") != "PeerHeadGather":
        bad.append("peer mapping failed: " + str(peer.get("exchange")))
    print(f"{cfg} N={n}: value {d['value']:.0f} {d['unit']} ({d['ms_per_step']:.3f} ms/step, n_gpus {d['n_gpus']}), "
          f"per-shard selection {d.get('value_per_shard_selection')}, shard_parity {sh.get('shard_parity')}, "
          + ", ".join(f"{m}: eager {v.get('eager_tokens_per_s', 0):.1f} / graph {v.get('graph_replay_tokens_per_s', float('nan')):.1f} tok/s"
                      for m, v in modes.items()))
else:
    print(f"{cfg} N=1: value {d['value']:.0f} {d['unit']} ({d['ms_per_step']:.3f} ms/step), decode {d.get('decode', {}).get('tokens_per_s')}")
for b in bad:
    print(f"FAIL {cfg} N={n}: {b}")
PY
  done
done
grep -q "^FAIL" $OUT/summary.txt && FAIL=1
echo "preflight: $([ $FAIL = 0 ] && echo PASSED || echo FAILED)  (summary: $OUT/summary.txt)"
exit $FAIL
