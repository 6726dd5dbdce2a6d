mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -p no:cacheprovider -k "gemm" 2>&1 | tail -3
for shp in "3992 1920 1920 0 0" "3992 1920 7680 0 0" "3992 1920 7680 0 1" "3992 1920 5760 0 1" "3992 1920 1920 0 1" "3992 5760 1920 0 0" "3992 7680 1920 0 0"; do
  for epi in 0 2; do
    for force in 1 2 3; do
      python tools/dev_gemm_perf.py $shp 30 $force 0 0 $epi 2>/dev/null | tail -1
    done
  done
done
