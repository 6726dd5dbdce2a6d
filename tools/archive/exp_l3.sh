timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -p no:cacheprovider -k "gemm" 2>&1 | tail -2
for shp in "3992 1920 1920 0 0" "3992 1920 7680 0 0" "3992 1920 7680 0 1" "3992 5760 1920 0 0"; do
  for force in 1 2; do python tools/dev_gemm_perf.py $shp 30 $force 0 0 2 2>/dev/null | tail -1; done
done
for i in 1 2 3; do
  for cfg in "CA_GEMM_PREFER_L=0" "CA_GEMM_PREFER_L=1" "CA_GEMM_PREFER_L=3"; do
    env $cfg python bench.py --no-cpu-baseline --no-also --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$cfg', d['ms_per_step'], d['value'], r['all_gemm_tflops'], d['config']['loss'])"
  done
done
