#!/bin/bash
# Round-5 profile session (one gpurun call): the default bench's kernel stats / traffic / SQ counters, the steady-state
# per-step tables of the headline step and of the reference's two production recipes (R/makefile:79-137: wav2vec2-small
# and whisper-large at per_device_batch_size=64), and the B = 64 decode kernel stats.
bash tools/profile_round.sh r05 > gpurun_out/profile_r05_round.log 2>&1
bash tools/profile_steady.sh r05 > gpurun_out/profile_r05_steady.log 2>&1
bash tools/profile_steady.sh r05b64small --model wav2vec2-small --batch 64 >> gpurun_out/profile_r05_steady.log 2>&1
bash tools/profile_steady.sh r05large_b64 --model whisper-large --batch 64 >> gpurun_out/profile_r05_steady.log 2>&1
bash tools/profile_decode64.sh > gpurun_out/profile_r05_decode64.log 2>&1
python bench.py --model whisper-large --batch 64 --steps 4 --warmup 2 2>&1 | tail -1 | cut -c1-400
tail -3 gpurun_out/profile_r05_round.log | cut -c1-300; tail -12 gpurun_out/profile_r05_steady.log | cut -c1-240
du -sh gpurun_out | tail -1
