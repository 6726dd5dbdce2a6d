# Round 6: per-(kernel, grid) times of the captured per-token decode step.  usage: bash tools/r06/profile_token.sh <batch> <tag> [model]
B=${1:-16}; TAG=${2:-base}; MODEL=${3:-whisper-medium}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HIP_FORCE_DEV_KERNARG=1
OUT=gpurun_out/prof_r06_token${B}_${TAG}
rm -rf $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 tools/r06/token_step_time.py $MODEL $B > $OUT.log 2>&1
tail -1 $OUT.log
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 tools/archive/dev_trace_by_grid.py $f 0.5 > gpurun_out/r06_token${B}_${TAG}_by_grid.txt
head -24 gpurun_out/r06_token${B}_${TAG}_by_grid.txt
rm -rf $OUT
