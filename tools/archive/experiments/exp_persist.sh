mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_w2v2_gpu.py tests/test_depth_gpu.py tests/test_fullsize_gpu.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -2
for shp in "3992 7680 1920 0 0" "3992 7680 1920 0 1" "7680 1920 3992 1 1" "1920 7680 3992 1 1"; do
  for p in 0 1; do
    CA_X_PERSIST=$p python tools/dev_gemm_perf.py $shp 30 3 $( [ "${shp: -3}" = "1 1" ] && echo 1 || echo 0 ) 2>/dev/null | tail -1 | sed "s/^/persist=$p /"
  done
done
for p in 0 1; do CA_X_PERSIST=$p python tools/dev_gemm_perf.py 3992 7680 1920 0 0 30 3 0 0 1 2>/dev/null | tail -1 | sed "s/^/persist=$p /"; done
for p in 0 1; do CA_X_PERSIST=$p python tools/dev_gemm_perf.py 3992 7680 1920 0 1 30 3 0 0 3 2>/dev/null | tail -1 | sed "s/^/persist=$p /"; done
for i in 1 2 3; do
  for p in 0 1; do
    CA_X_PERSIST=$p python bench.py --no-cpu-baseline --no-also --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('persist=$p', d['ms_per_step'], d['value'], r['all_gemm_tflops'], r['kernel'][:36], r['achieved'], d['config']['loss'])"
  done
done
