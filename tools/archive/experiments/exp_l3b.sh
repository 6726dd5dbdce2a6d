for i in 1 2; do
  for cfg in "CA_GEMM_PREFER_L=3" "CA_GEMM_PREFER_L=8" "CA_GEMM_PREFER_L=8 CA_GEMM_L_OVER_X=1" "CA_GEMM_PREFER_L=8 CA_GEMM_L_OVER_X=1 CA_FUSE_BIAS=0" "CA_GEMM_PREFER_L=8 CA_GEMM_L_OVER_X=2 CA_FUSE_BIAS=0"; do
    env $cfg python bench.py --no-cpu-baseline --no-also --steps 10 --warmup 3 --gemm-breakdown 2>/tmp/gb.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$cfg', d['ms_per_step'], d['value'], r['all_gemm_tflops'], d['config']['loss'])"
    [ $i = 1 ] && grep TFLOP /tmp/gb.txt | head -6
  done
done
