cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HIP_FORCE_DEV_KERNARG=1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r05_decode64 -- python3 bench.py --model whisper-medium --decode --batch 64 --steps 3 --warmup 1 > gpurun_out/prof_r05_decode64.log 2>&1
f=$(find gpurun_out/prof_r05_decode64 -name "*kernel_stats.csv" | head -1); head -14 $f | cut -c1-220
find gpurun_out/prof_r05_decode64 -name "*kernel_trace.csv" -delete; find gpurun_out/prof_r05_decode64 -name "*.db" -delete
