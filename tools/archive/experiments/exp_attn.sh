mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_w2v2_gpu.py tests/test_whisper_gpu.py tests/test_depth_gpu.py tests/test_fullsize_gpu.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CA_WGRAD_STREAM=0
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tmp_stats -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-also > gpurun_out/prof_tmp_bench.log 2>&1
find gpurun_out -name "*kernel_trace.csv" -delete
f=$(ls gpurun_out/prof_tmp_stats/*/*kernel_stats.csv | head -1)
grep -E "attn_|ln_|Name" $f | cut -d, -f1-4 | cut -c1-140
unset CA_WGRAD_STREAM
for i in 1 2; do python bench.py --no-cpu-baseline --no-also --steps 10 2>/dev/null | tail -1 | cut -c1-330; done
