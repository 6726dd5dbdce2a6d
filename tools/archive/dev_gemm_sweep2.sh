#!/bin/bash
# epilogue cost of the fused FFN GEMMs on kernel S (1) and X (3)
for f in 1 3; do
  python tools/dev_gemm_perf.py 3992 7680 1920 0 0 20 $f 0 0 0
  python tools/dev_gemm_perf.py 3992 7680 1920 0 0 20 $f 0 0 1
  python tools/dev_gemm_perf.py 3992 7680 1920 0 1 20 $f 0 0 0
  python tools/dev_gemm_perf.py 3992 7680 1920 0 1 20 $f 0 0 3
  python tools/dev_gemm_perf.py 3992 1920 7680 0 0 20 $f 0 0 2
  python tools/dev_gemm_perf.py 3992 1920 1920 0 0 20 $f 0 0 0
  python tools/dev_gemm_perf.py 3992 5760 1920 0 0 20 $f 0 0 0
  python tools/dev_gemm_perf.py 1920 1920 3992 1 1 20 $f 1 0 0
  python tools/dev_gemm_perf.py 5760 1920 3992 1 1 20 $f 1 0 0
done
