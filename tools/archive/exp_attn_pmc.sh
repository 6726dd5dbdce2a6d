#!/bin/bash
# SQ counters of the attention kernels at the models' shapes (two --pmc passes over tools/dev_attn_perf.py), for the
# library in CORAL_AMD_LIB (default: the in-tree build).  bash tools/archive/exp_attn_pmc.sh [tag]
cd /tmp; export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1; R=$GRAFT_REPO_ROOT; T=${1:-new}
rm -rf /tmp/p1_$T /tmp/p2_$T
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d /tmp/p1_$T -- python3 $R/tools/dev_attn_perf.py > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d /tmp/p2_$T -- python3 $R/tools/dev_attn_perf.py > /dev/null 2>&1
python3 $R/tools/archive/dev_attn_counters.py $(find /tmp/p1_$T /tmp/p2_$T -name "*counter_collection.csv") --filter "attn"
