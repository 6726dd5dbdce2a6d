#!/bin/bash
# dK|dV kernel: 64-key workgroups (a wave owns 16 keys) against 128-key workgroups (a wave owns 2 x 16 keys), per-kernel
# times from rocprofv3 at the models' shapes.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in 0 1; do
  export CA_ATTN_DKV_WIDE=$w
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dkv$w -- python tools/dev_attn_perf.py > gpurun_out/dkv$w.log 2>&1
  echo "== CA_ATTN_DKV_WIDE=$w"; grep -v "^[WEI][0-9]\|amdgpu" gpurun_out/dkv$w.log | tail -6
  python - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof_dkv$w/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "attn_bwd_dkv" in r["Name"]:
        print(r["Name"][:60], r["Calls"], "avg us", float(r["AverageNs"]) / 1e3, "min", float(r["MinNs"]) / 1e3, "max", float(r["MaxNs"]) / 1e3)
PY
done
find gpurun_out -name "*kernel_trace.csv" -delete
