#!/bin/bash
# Round 6: fabric / HBM traffic of the persistent decode launch (separate --pmc passes, as the microarch guide prescribes):
#   bash tools/r06/profile_decode_traffic.sh [clips]   ->  gpurun_out/r06_decode_traffic_<clips>.txt
B=${1:-16}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HIP_FORCE_DEV_KERNARG=1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/prof_r06_dec_$c -- python tools/r06/token_step_time.py whisper-medium $B > gpurun_out/prof_r06_dec_$c.log 2>&1
done
python - $B <<'PY' > gpurun_out/r06_decode_traffic_$1.txt
import csv, glob, sys, collections
B = sys.argv[1]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/prof_r06_dec_{c}/*/*counter_collection.csv")
    vals = collections.defaultdict(list)
    for row in csv.DictReader(open(f[0])):
        if row["Counter_Name"] == c:
            vals[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    res[c] = vals
print(f"# whisper-medium, {B} clips: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KiB, separate passes) of tools/r06/token_step_time.py")
print("# bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB (gfx950: FETCH_SIZE counts half of a wide streaming read)")
for k in res["FETCH_SIZE"]:
    if "decode_token" not in k and "smallq" not in k and "skinny" not in k:
        continue
    fv, wv = res["FETCH_SIZE"][k], res["WRITE_SIZE"].get(k, [0.0])
    n = len(fv)
    fa, wa = sum(fv) / n, sum(wv) / max(1, len(wv))
    print(f"{k[:90]:90s} launches {n:5d}  FETCH_SIZE {fa:12.1f} KiB  WRITE_SIZE {wa:10.1f} KiB  -> {(2 * fa + wa) * 1024 / 1e9:7.3f} GB per launch")
PY
cat gpurun_out/r06_decode_traffic_$1.txt
find gpurun_out/prof_r06_dec_* -name "*.db" -delete
