#!/bin/bash
# Same-box A/B of values of one environment variable on the bench step (interleaved):
#   bash tools/exp_envval_ab.sh GPU_MAX_HW_QUEUES "4 8 2" [bench args...]  -> whole step | forward+backward | GEMM ms per step
V=$1
VALS=$2
ARGS="${@:3}"
show() { python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('$1', d['ms_per_step'], (c.get('fwd_bwd') or {}).get('ms_per_step'), d.get('roofline',{}).get('gemm_ms_per_step'))"; }
for i in 1 2; do
  for x in $VALS; do
    env $V=$x python bench.py --no-cpu-baseline --no-also --steps 10 --warmup 3 $ARGS 2>/dev/null | tail -1 | show "$V=$x"
  done
done
