for i in 1 2 3; do
  for cfg in "CA_GEMM_L_OVER_X=0" "CA_GEMM_L_OVER_X=3"; do
    env $cfg python bench.py --no-cpu-baseline --no-also --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$cfg', d['ms_per_step'], d['value'], r['all_gemm_tflops'], d['config']['loss'])"
  done
done
for cfg in "CA_OPT_OVERLAP=0 CA_WGRAD_STREAM=0" "CA_OPT_OVERLAP=1 CA_WGRAD_STREAM=0" "CA_OPT_OVERLAP=1 CA_WGRAD_STREAM=1"; do env $cfg python bench.py --no-cpu-baseline --no-also --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(\"$cfg\", d[\"ms_per_step\"])"; done
