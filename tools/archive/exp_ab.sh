# (_base/ = a worktree of the round-start commit, built as-is: git worktree add _base b11f60a && make -C _base/coral_amd/csrc; not kept in the tree)
# same-box A/B: round-start tree (_base/, commit b11f60a built as-is) against the working tree, interleaved
for i in 1 2 3; do
  (cd _base && python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('base', d['ms_per_step'], d['value'])")
  python bench.py --no-cpu-baseline --no-also --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new ', d['ms_per_step'], d['value'])"
done
(cd _base && python bench.py --no-cpu-baseline --model wav2vec2-small --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('base small', d['ms_per_step'], d['value'])")
python bench.py --no-cpu-baseline --no-also --model wav2vec2-small --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new  small', d['ms_per_step'], d['value'])"
