#!/bin/bash
# Same-box A/B of one environment switch on the bench step (interleaved pairs):
#   bash tools/exp_env_ab.sh CA_STREAM_WGRAD [bench args...]      -> whole step | forward+backward | GEMM ms per step
V=$1
ARGS="${@:2}"
show() { python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('$1', d['ms_per_step'], (c.get('fwd_bwd') or {}).get('ms_per_step'), d.get('roofline',{}).get('gemm_ms_per_step'))"; }
for i in 1 2 3; do
  env $V=0 python bench.py --no-cpu-baseline --no-also --steps 10 --warmup 3 $ARGS 2>/dev/null | tail -1 | show "$V=0"
  env $V=1 python bench.py --no-cpu-baseline --no-also --steps 10 --warmup 3 $ARGS 2>/dev/null | tail -1 | show "$V=1"
done
