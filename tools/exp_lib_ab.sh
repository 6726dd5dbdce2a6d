#!/bin/bash
# Same-box A/B of two builds of the library (interleaved pairs):  bash tools/exp_lib_ab.sh <variant.so> [bench args...]
# A = coral_amd/libcoral_amd.so, B = the variant (CORAL_AMD_LIB).  Prints whole step and forward+backward ms per run.
VAR=$1
ARGS="${@:2}"
show() { python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('$1', d['ms_per_step'], (c.get('fwd_bwd') or {}).get('ms_per_step'), d.get('roofline',{}).get('gemm_ms_per_step'))"; }
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --no-also --steps 10 --warmup 3 $ARGS 2>/dev/null | tail -1 | show "A"
  CORAL_AMD_LIB=$PWD/$VAR python bench.py --no-cpu-baseline --no-also --steps 10 --warmup 3 $ARGS 2>/dev/null | tail -1 | show "B"
done
