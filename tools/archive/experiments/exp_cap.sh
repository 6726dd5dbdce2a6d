for i in 1 2; do
  for cfg in "CA_ADAMW_BLOCKS=0" "CA_ADAMW_BLOCKS=2048" "CA_ADAMW_BLOCKS=1024" "CA_ADAMW_BLOCKS=768" "CA_ADAMW_BLOCKS=512"; do
    env $cfg python bench.py --no-cpu-baseline --no-also --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', d['ms_per_step'], d['value'])"
  done
done
