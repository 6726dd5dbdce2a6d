#!/bin/bash
# Round 6: SQ counters of the persistent decode launch (own --pmc pass):  bash tools/r06/profile_decode_sq.sh [clips]
B=${1:-16}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HIP_FORCE_DEV_KERNARG=1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof_r06_dec_sq -- python tools/r06/token_step_time.py whisper-medium $B > gpurun_out/prof_r06_dec_sq.log 2>&1
python - $B <<'PY' > gpurun_out/r06_decode_sq_$1.txt
import csv, glob, sys, collections
B = sys.argv[1]
f = glob.glob("gpurun_out/prof_r06_dec_sq/*/*counter_collection.csv")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(f[0])):
    acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
print(f"# whisper-medium, {B} clips: rocprofv3 --pmc SQ_* of tools/r06/token_step_time.py, averages per launch (chip-wide sums)")
for k, cs in acc.items():
    if "decode_token" not in k:
        continue
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    n = len(next(iter(cs.values())))
    print(k[:80], "launches", n)
    for c in sorted(m):
        print(f"  {c:22s} {m[c]:16.0f}")
    if m.get("SQ_WAVE_CYCLES"):
        wc = 4.0 * m["SQ_WAVE_CYCLES"]  # (SQ_WAVE_CYCLES counts quad-cycles: 4 x = wave-cycles summed over the waves)
        print(f"  one VALU instruction per {wc / max(1.0, m.get('SQ_INSTS_VALU', 1)):.1f} cycles of a wave, one MFMA per "
              f"{m.get('SQ_INSTS_VALU', 0) / max(1.0, m.get('SQ_INSTS_MFMA', 1)):.0f} VALU; SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = "
              f"{m.get('SQ_WAIT_INST_ANY', 0) / m['SQ_WAVE_CYCLES']:.3f}")
PY
cat gpurun_out/r06_decode_sq_$1.txt
find gpurun_out/prof_r06_dec_sq -name "*.db" -delete
