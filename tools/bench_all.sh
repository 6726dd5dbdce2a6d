# every bench line quoted in DESIGN.md §5, one lease:  bash tools/bench_all.sh > gpurun_out/bench_all.log
B="python bench.py --no-cpu-baseline --no-also"
one() { "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:90], '|', d['ms_per_step'], 'ms |', d['value'], d['unit'])"; }
one $B
one $B
one $B --ragged
one $B --from-host-pcm
one $B --model wav2vec2-small
one $B --model wav2vec2-medium
one $B --model whisper-medium
one $B --model whisper-medium --decode
one $B --model whisper-large-turbo
one $B --model whisper-large-turbo --fp8-forward
