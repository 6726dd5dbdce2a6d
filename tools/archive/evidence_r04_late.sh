#!/bin/bash
# The same-box measurements behind DESIGN.md 5.0 "Round 4, late", in one lease -> gpurun_out/r04_late_evidence.txt
# (copied to profiles/r04_late_evidence.txt).
out=gpurun_out/r04_late_evidence.txt
mkdir -p gpurun_out
{
echo "# one MI355X box, one lease; every block is interleaved A | B on this box"
echo "## 1. kernel arguments in device memory: bench.py --steps 10 --warmup 3 (whole step ms | forward+backward ms | GEMM ms per step)"
bash tools/exp_envval_ab.sh HIP_FORCE_DEV_KERNARG "0 1"
echo "## 2. AdamW under the next forward: events on the main and optimiser streams (tools/archive/dev_opt_timeline.py), ms"
for s in "CA_OPT_OVERLAP=0" "CA_OPT_BG_BLOCKS=0" "CA_OPT_BG_BLOCKS=256" "CA_OPT_BG_BLOCKS=128" "CA_OPT_BG_BLOCKS=512"; do
  echo "$s : $(env $s python tools/archive/dev_opt_timeline.py 2>&1 | grep '^forward')"
done
echo "## 3. whole step with the background optimiser off | on"
bash tools/exp_envval_ab.sh CA_OPT_BG_BLOCKS "0 256"
echo "## 4. one side stream per role and process: whisper-large-turbo fp8 / bf16 engines built one after another in ONE process (ms per step)"
echo "-- CA_SHARED_STREAMS=0 (a stream per engine)"; CA_SHARED_STREAMS=0 python tools/archive/dev_turbo_inproc.py bfbf 0 gc 2>&1 | grep "^fp8\|^bf16"
echo "-- CA_SHARED_STREAMS=1 (default)";               python tools/archive/dev_turbo_inproc.py bfbf 0 gc 2>&1 | grep "^fp8\|^bf16"
echo "## 5. decode cross-attention kernels alone (tools/archive/dev_cross_attn_time.py), us per launch"
python tools/archive/dev_cross_attn_time.py 4 8 16 32 2>&1 | grep "^B="
echo "## 6. greedy decoding, whisper-medium, ms per 36-token pass (8 | 16 clips): LayerNorm in the projection prologue off | on, key split off | on"
for v in "CA_DECODE_LN_FUSED=0 CA_ATTN_SPLIT=1" "CA_DECODE_LN_FUSED=1 CA_ATTN_SPLIT=1" "CA_DECODE_LN_FUSED=1 CA_ATTN_SPLIT=4"; do
  for B in 8 16; do echo "$v B=$B $(env $v python bench.py --model whisper-medium --decode --batch $B --steps 12 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms per pass;', d['config'].get('ms_per_token'), 'ms per token')")"; done
done
} > $out 2>&1
tail -5 $out
