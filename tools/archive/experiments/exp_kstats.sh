# usage: bash tools/exp_kstats.sh '<grep -E pattern of kernel names>'  -> GPU tests (fail-fast), per-kernel averages, bench lines
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -m gpu -x -p no:cacheprovider 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CA_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tmp_stats -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-also > gpurun_out/prof_tmp_bench.log 2>&1
find gpurun_out -name "*kernel_trace.csv" -delete
f=$(ls -t gpurun_out/prof_tmp_stats/*/*kernel_stats.csv | head -1)
grep -E "$1" $f | awk -F'",' '{split($2,a,","); printf "%-70s calls %6d avg_us %8.1f total_ms/step %7.2f\n", substr($1,2,68), a[1], a[3]/1e3, a[2]/9e6}'
for i in 1 2; do python bench.py --no-cpu-baseline --no-also --steps 10 2>/dev/null | tail -1 | cut -c100-220; done
