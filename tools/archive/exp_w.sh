#!/bin/bash
# Kernel W (4 waves x 128x128, one wave per SIMD) against kernel X (8 waves x 128x64): steady state and the path's shapes.
cd "$(dirname "$0")/.."
echo "== steady state, K=16384, force X (3) =="; python tools/archive/dev_gemm_steady.py 16384 3
echo "== steady state, K=16384, force W (4) =="; python tools/archive/dev_gemm_steady.py 16384 4
for rep in 1 2; do
for f in 3 4; do
  echo "-- force $f (rep $rep)"
  python tools/dev_gemm_perf.py 3992 7680 1920 0 0 30 $f      # fc1 forward (plain)
  python tools/dev_gemm_perf.py 3992 7680 1920 0 0 30 $f 0 0 1  # fc1 forward, GELU + dropout, two outputs
  python tools/dev_gemm_perf.py 3992 7680 1920 0 1 30 $f 0 0 3  # fc2 dgrad with GELU'
  python tools/dev_gemm_perf.py 7680 1920 3992 1 1 30 $f 1      # weight gradient, fp32 +=
  python tools/dev_gemm_perf.py 3992 1920 7680 0 0 30 $f      # fc2 forward
  python tools/dev_gemm_perf.py 3992 5760 1920 0 0 30 $f      # q|k|v
done
done
