mkdir -p gpurun_out
python -c "import torch; print('prio range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')" > gpurun_out/exp1.log 2>&1
B="python bench.py --no-cpu-baseline --no-also --steps 10 --warmup 3"
for cfg in "" "CA_OPT_PRIO=1" "CA_ADAMW_BLOCKS=512" "CA_ADAMW_BLOCKS=256" "CA_ADAMW_BLOCKS=128" "CA_ADAMW_BLOCKS=256 CA_OPT_PRIO=1" ""; do
  echo "== $cfg" >> gpurun_out/exp1.log
  env $cfg $B 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(d['ms_per_step'], d['value'], r['all_gemm_tflops'], r['gemm_ms_per_step'])" >> gpurun_out/exp1.log
done
python tools/dev_gemm_shapes.py > gpurun_out/shapes_noopt.log 2>&1
cat gpurun_out/exp1.log; tail -30 gpurun_out/shapes_noopt.log
