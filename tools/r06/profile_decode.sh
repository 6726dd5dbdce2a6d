# Round 6: per-kernel profile of greedy decoding at a given batch.  usage: bash tools/r06/profile_decode.sh <batch> <tag>
B=${1:-16}; TAG=${2:-base}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HIP_FORCE_DEV_KERNARG=1
OUT=gpurun_out/prof_r06_decode${B}_${TAG}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --model whisper-medium --decode --batch $B --steps 3 --warmup 1 --no-also --no-cpu-baseline > $OUT.log 2>&1
tail -1 $OUT.log | cut -c1-400
f=$(find $OUT -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r06_decode${B}_${TAG}_kernel_stats.csv; head -12 $f | cut -c1-200
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
