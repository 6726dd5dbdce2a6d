#!/bin/bash
# Round profile on the GPU box (one gpurun call):  bash tools/profile_round.sh r02
#  1. tools/profile_bench.sh: kernel-trace stats of the default bench command + FETCH_SIZE / WRITE_SIZE passes
#  2. SQ counter pass of the same command (own run, --pmc only): MFMA-pipe busy cycles, VALU / MFMA instruction counts
#  3. kernel-trace stats of the Whisper greedy-decode bench
# tools/summarize_profile.py <tag> then writes profiles/<tag>_*.csv and profiles/pmc_traffic.json.
TAG=${1:-r02}
bash tools/profile_bench.sh $TAG
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# per-kernel durations and counters need serialised kernels: weight gradients and the optimiser back on the main stream
# (see bench.py)
export CA_WGRAD_STREAM=0 CA_OPT_OVERLAP=0
# (bench.py sets this itself, but a profiler that initialises the runtime first would read the environment before it does)
export HIP_FORCE_DEV_KERNARG=1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof_${TAG}_sq -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-also --no-fwd-bwd > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_decode_stats -- python bench.py --model whisper-medium --decode --steps 4 --warmup 2 > gpurun_out/prof_${TAG}_decode.log 2>&1
tail -1 gpurun_out/prof_${TAG}_decode.log | cut -c1-300
# the per-dispatch traces are large (gpurun copies back at most 64 MiB): keep the summaries only
find gpurun_out -name "*kernel_trace.csv" -delete
find gpurun_out -name "*.db" -delete
du -sh gpurun_out/prof_${TAG}_* | head -20
grep -v "^[WEI][0-9]" gpurun_out/prof_${TAG}_bench.log | tail -5 | cut -c1-400
