#!/bin/bash
# Kernel-trace stats of the secondary workloads (one gpurun call): the 300M shape (BASELINE configs[0]) and the whisper-medium
# finetune step (configs[3]).  tools/summarize_profile.py <tag>small / <tag>whisper writes profiles/<tag>*_kernel_stats.csv.
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CA_WGRAD_STREAM=0 CA_OPT_OVERLAP=0
# (bench.py sets this itself, but a profiler that initialises the runtime first would read the environment before it does)
export HIP_FORCE_DEV_KERNARG=1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}small_stats -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-also --no-fwd-bwd --model wav2vec2-small > gpurun_out/prof_${TAG}small_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}whisper_stats -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-also --no-fwd-bwd --model whisper-medium > gpurun_out/prof_${TAG}whisper_bench.log 2>&1
find gpurun_out -name "*kernel_trace.csv" -delete
find gpurun_out -name "*.db" -delete
grep -v "^[WEI][0-9]" gpurun_out/prof_${TAG}small_bench.log | tail -1 | cut -c1-300
grep -v "^[WEI][0-9]" gpurun_out/prof_${TAG}whisper_bench.log | tail -1 | cut -c1-300
