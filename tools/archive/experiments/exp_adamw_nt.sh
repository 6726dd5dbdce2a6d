for i in 1 2 3; do
  for cfg in "CA_ADAMW_NT=0" "CA_ADAMW_NT=1"; do
    env $cfg python bench.py --no-cpu-baseline --no-also --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$cfg', d['ms_per_step'], d['value'], r['all_gemm_tflops'], r['gemm_ms_per_step'], r['kernel'][:40], r['achieved'])"
  done
done
