#!/bin/bash
# Kernel choice per GEMM of the XLS-R-2B step after the round-3 epilogue: S (1) / L (2) / X (3) / auto (0), same box.
cd "$(dirname "$0")/.."
run() { for f in 0 1 2 3; do python tools/dev_gemm_perf.py "$@" 2>&1 | grep -v amdgpu | sed "s/^/force-arg /" ; done; }
for f in 0 1 2 3; do
  echo "== force $f"
  python tools/dev_gemm_perf.py 3992 5760 1920 0 0 30 $f 0 0 0 2>&1 | grep -v amdgpu   # q|k|v
  python tools/dev_gemm_perf.py 3992 1920 1920 0 0 30 $f 0 0 2 2>&1 | grep -v amdgpu   # out-projection + residual
  python tools/dev_gemm_perf.py 3992 7680 1920 0 0 30 $f 0 0 1 2>&1 | grep -v amdgpu   # fc1 + GELU + dropout
  python tools/dev_gemm_perf.py 3992 1920 7680 0 0 30 $f 0 0 2 2>&1 | grep -v amdgpu   # fc2 + residual
  python tools/dev_gemm_perf.py 3992 7680 1920 0 1 30 $f 0 0 3 2>&1 | grep -v amdgpu   # fc2 dgrad + GELU'
  python tools/dev_gemm_perf.py 3992 1920 7680 0 1 30 $f 0 0 0 2>&1 | grep -v amdgpu   # fc1 dgrad
  python tools/dev_gemm_perf.py 3992 1920 1920 0 1 30 $f 0 0 0 2>&1 | grep -v amdgpu   # out dgrad
  python tools/dev_gemm_perf.py 3992 1920 5760 0 1 30 $f 0 0 0 2>&1 | grep -v amdgpu   # q|k|v dgrad
  python tools/dev_gemm_perf.py 7680 1920 3992 1 1 30 $f 1 2>&1 | grep -v amdgpu       # fc1 wgrad
  python tools/dev_gemm_perf.py 1920 7680 3992 1 1 30 $f 1 2>&1 | grep -v amdgpu       # fc2 wgrad
  python tools/dev_gemm_perf.py 5760 1920 3992 1 1 30 $f 1 2>&1 | grep -v amdgpu       # q|k|v wgrad
done
