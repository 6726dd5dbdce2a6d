#!/bin/bash
# usage: tools/dev_gemm_sweep.sh "<force list>"   (force = kernel + 16*variant)
for f in $1; do
  python tools/dev_gemm_perf.py 7680 1920 3992 1 1 20 $f 1
  python tools/dev_gemm_perf.py 3992 7680 1920 0 0 20 $f 0
  python tools/dev_gemm_perf.py 3992 7680 1920 0 1 20 $f 0
done
