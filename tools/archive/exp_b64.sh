#!/bin/bash
# Kernel choice at the production batch (XLS-R-300M, 64 x 10 s: M = 31936 tokens, d 1024, ffn 4096): S / L / X per shape.
cd "$(dirname "$0")/.."
for shape in "31936 4096 1024 0 0 0 0 0 1" "31936 1024 4096 0 0 0 0 0 2" "31936 3072 1024 0 0" "31936 1024 1024 0 0 0 0 0 2" \
             "31936 4096 1024 0 1 0 0 0 3" "31936 1024 4096 0 1" "31936 1024 3072 0 1" "31936 1024 1024 0 1"; do
  set -- $shape
  for f in 1 2 3; do
    python tools/dev_gemm_perf.py $1 $2 $3 $4 $5 20 $f ${6:-0} ${7:-0} ${9:-0} 2>&1 | grep -v amdgpu
  done
done
