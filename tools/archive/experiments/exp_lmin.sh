for m in wav2vec2-small whisper-medium; do
for i in 1 2; do
  for cfg in "CA_GEMM_L_MIN=160" "CA_GEMM_L_MIN=100" "CA_GEMM_L_MIN=48"; do
    env $cfg python bench.py --no-cpu-baseline --no-also --model $m --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$m $cfg', d['ms_per_step'], d['value'])"
  done
done
done
