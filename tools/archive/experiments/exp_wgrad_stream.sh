mkdir -p gpurun_out
L=gpurun_out/exp_wgrad.log; : > $L
timeout 900 python -m pytest tests/test_w2v2_gpu.py tests/test_depth_gpu.py tests/test_dp_gpu.py tests/test_fullsize_gpu.py tests/test_finetune_gpu.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -5 >> $L
B="python bench.py --no-cpu-baseline --no-also --steps 10 --warmup 3"
for cfg in "CA_WGRAD_STREAM=0" "CA_WGRAD_STREAM=1" "CA_WGRAD_STREAM=1 CA_WGRAD_PRIO=-1" "CA_WGRAD_STREAM=0" "CA_WGRAD_STREAM=1"; do
  echo "== $cfg" >> $L
  env $cfg $B 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(d['ms_per_step'], d['value'], r['all_gemm_tflops'], r['gemm_ms_per_step'], d['config']['loss'])" >> $L
done
cat $L
