#!/bin/bash
# Steady-state per-step kernel profile on the GPU box:  bash tools/profile_steady.sh <tag> [bench.py args...]
# Two kernel-trace passes of the same command with 2 and 6 timed steps; tools/summarize_steady.py <tag> takes the
# DIFFERENCE (calls and time per kernel, divided by the 4 extra steps), so set-up work - parameter initialisation, the
# workspace arena's fill, warm-up - cancels and profiles/<tag>_per_step.csv shows exactly what ONE step launches.
TAG=$1
ARGS="${@:2}"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CA_WGRAD_STREAM=0 CA_OPT_OVERLAP=0
# (bench.py sets this itself, but a profiler that initialises the runtime first would read the environment before it does)
export HIP_FORCE_DEV_KERNARG=1
for n in 2 6; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_steady$n -- python bench.py --steps $n --warmup 2 --no-cpu-baseline --no-also --no-fwd-bwd $ARGS > gpurun_out/prof_${TAG}_steady$n.log 2>&1
  grep -v "^[WEI][0-9]" gpurun_out/prof_${TAG}_steady$n.log | tail -1 | cut -c1-200
done
find gpurun_out -name "*kernel_trace.csv" -delete
find gpurun_out -name "*.db" -delete
