#!/bin/bash
# Round-6 profile session (one gpurun call): the default bench's kernel stats / traffic / SQ counters (profile_round.sh: its
# decode pass now profiles the persistent launch), the steady-state per-step table of the headline step, the default bench line.
bash tools/profile_round.sh r06 > gpurun_out/profile_r06_round.log 2>&1
bash tools/profile_steady.sh r06 > gpurun_out/profile_r06_steady.log 2>&1
python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
tail -c 1500 gpurun_out/r06_bench_default.json
tail -3 gpurun_out/profile_r06_round.log | cut -c1-300; tail -4 gpurun_out/profile_r06_steady.log | cut -c1-240
du -sh gpurun_out | tail -1
