#!/bin/bash
# Same-box comparison of builds of the library on the FFN launch shapes (kernel X forced), interleaved, two rounds:
#   bash tools/exp_r05_lib_abc.sh base v1 cur     (coral_amd/libcoral_amd_<name>.so; "cur" = coral_amd/libcoral_amd.so)
for round in 1 2; do
for v in "$@"; do
  lib=$PWD/coral_amd/libcoral_amd_$v.so; [ "$v" = cur ] && lib=$PWD/coral_amd/libcoral_amd.so
  for cfg in "3992 7680 1920 0 0 20 3 0 0 0" "3992 7680 1920 0 0 20 3 0 0 1" "3992 7680 1920 0 1 20 3 0 0 3" "12000 5120 1280 0 0 20 3 0 0 1" "31936 4096 1024 0 0 20 3 0 0 0" "31936 4096 1024 0 0 20 3 0 0 1" "7680 1920 3992 1 1 20 3 1 0 0"; do
    echo -n "$v r$round: "; CORAL_AMD_LIB=$lib python tools/dev_gemm_perf.py $cfg 2>&1 | tail -1
  done
done; done
