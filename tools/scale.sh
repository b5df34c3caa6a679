#!/bin/bash
# The 1 -> 8 GPU scaling curve of one node, one command:
#   tools/scale.sh [--configs "c3 c5"] [--gpus "1 2 4 8"] [--steps K] [--warmup W] [--out DIR] [--port P] [--dry-run]
# For every configuration and every N it runs the driver's own launch line
#   N = 1:  python bench.py --config C --gpus 1 ...
#   N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --config C --gpus N ...
# (weak scaling: 1024 env instances per GPU on c3 = 8192 at N = 8, BASELINE.json configs[3]; 512 per GPU on c5 = 4096, configs[4]),
# keeps the one JSON line of each run in DIR/scale_<config>_<N>.json and then CHECKS every line: n_gpus == N, and for N > 1
# extra.ranks.backend names nccl (= RCCL) and extra.ranks.distinct_devices == N -- a curve whose ranks shared a device, or ran over
# gloo, fails here instead of being reported.  --dry-run prints the commands and checks nothing (what the CPU test exercises).
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CONFIGS="c3 c5"; GPUS="1 2 4 8"; STEPS=""; WARMUP=""; OUT="$ROOT/gpurun_out/scale"; PORT=29533; DRY=0
while [ $# -gt 0 ]; do
  case "$1" in
    --configs) CONFIGS="$2"; shift 2;;
    --gpus) GPUS="$2"; shift 2;;
    --steps) STEPS="$2"; shift 2;;
    --warmup) WARMUP="$2"; shift 2;;
    --out) OUT="$2"; shift 2;;
    --port) PORT="$2"; shift 2;;
    --dry-run) DRY=1; shift;;
    -h|--help) sed -n 2,12p "$0"; exit 0;;
    *) echo "scale.sh: unknown argument '$1'" >&2; exit 2;;
  esac
done
for C in $CONFIGS; do
  case $C in c2|c3|c5|q1) ;; *) echo "scale.sh: unknown config '$C' (c2 c3 c5 q1)" >&2; exit 2;; esac
done
for N in $GPUS; do
  case $N in 1|2|4|8) ;; *) echo "scale.sh: --gpus takes 1, 2, 4 or 8 (one node), not '$N'" >&2; exit 2;; esac
done
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p "$OUT"
FAIL=0
for C in $CONFIGS; do
  for N in $GPUS; do
    ARGS="--config $C --gpus $N --no-cpu-baseline --no-extra"
    [ -n "$STEPS" ] && ARGS="$ARGS --steps $STEPS"
    [ -n "$WARMUP" ] && ARGS="$ARGS --warmup $WARMUP"
    if [ "$N" = 1 ]; then
      CMD="python $ROOT/bench.py $ARGS"
    else
      CMD="python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT $ROOT/bench.py $ARGS"
    fi
    F="$OUT/scale_${C}_${N}.json"
    if [ $DRY = 1 ]; then echo "$CMD > $F"; continue; fi
    echo "[scale] $CMD" >&2
    $CMD > "$F" 2> "$OUT/scale_${C}_${N}.err" || { echo "[scale] $C N=$N: the run failed (see $OUT/scale_${C}_${N}.err)" >&2; FAIL=1; continue; }
    python - "$F" "$N" <<'PY' || FAIL=1
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith('{')][-1]
d, n = json.loads(line), int(sys.argv[2])
r = d['extra']['ranks']
assert d['n_gpus'] == n, 'n_gpus %r != %d' % (d['n_gpus'], n)
if n > 1:
    assert str(r['backend']).startswith('nccl'), 'backend %r is not nccl (RCCL)' % r['backend']
    assert r['distinct_devices'] == n, '%d ranks on %d distinct devices' % (n, r['distinct_devices'])
print('%s N=%d: %.4g %s, %.2f ms per step, %s' % (d['metric'], n, d['value'], d['unit'], d['ms_per_step'], r['backend']))
PY
  done
done
[ $DRY = 1 ] || { echo "[scale] lines in $OUT"; }
exit $FAIL
