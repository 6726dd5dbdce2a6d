timeout 900 python -m pytest tests/test_w2v2_gpu.py tests/test_depth_gpu.py tests/test_dp_gpu.py tests/test_fullsize_gpu.py tests/test_finetune_gpu.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -2
for i in 1 2 3; do
  (cd _base && python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('base', d['ms_per_step'], d['value'])")
  python bench.py --no-cpu-baseline --no-also --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new ', d['ms_per_step'], d['value'])"
  CA_WGRAD_STREAM=0 python bench.py --no-cpu-baseline --no-also --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new, one stream', d['ms_per_step'], d['value'])"
done
