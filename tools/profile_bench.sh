#!/bin/bash
# Round profile of the default bench command on the GPU box: kernel stats + the PMC passes that give
# HBM traffic of every kernel (FETCH_SIZE / WRITE_SIZE in separate passes, as the microarch guide
# prescribes).  Outputs under gpurun_out/prof_$1; tools/summarize_profile.py turns them into profiles/.
TAG=${1:-r01}
ARGS="${@:2}"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# per-kernel durations and counters need serialised kernels: weight gradients and the optimiser back on the main stream
# (see bench.py)
export CA_WGRAD_STREAM=0 CA_OPT_OVERLAP=0
# (bench.py sets this itself, but a profiler that initialises the runtime first would read the environment before it does)
export HIP_FORCE_DEV_KERNARG=1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_stats -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-also --no-fwd-bwd $ARGS > gpurun_out/prof_${TAG}_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_${TAG}_fetch -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-also --no-fwd-bwd $ARGS > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_${TAG}_write -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-also --no-fwd-bwd $ARGS > /dev/null 2>&1
tail -1 gpurun_out/prof_${TAG}_bench.log | cut -c1-400
