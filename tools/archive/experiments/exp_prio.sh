for i in 1 2 3; do
  for cfg in "CA_MAIN_PRIO=0" "CA_MAIN_PRIO=-1"; do
    env $cfg python bench.py --no-cpu-baseline --no-also --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$cfg', d['ms_per_step'], d['value'], r['all_gemm_tflops'])"
  done
done
