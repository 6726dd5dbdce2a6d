cd /tmp; export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1; R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d /tmp/p1 -- python3 $R/tools/dev_attn_perf.py 2>&1 | grep -v "^[WEI][0-9]" | tail -8
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d /tmp/p2 -- python3 $R/tools/dev_attn_perf.py 2>&1 | grep -v "^[WEI][0-9]" | tail -8
python3 $R/tools/dev_attn_counters.py $(find /tmp/p1 /tmp/p2 -name "*counter_collection.csv") --filter "attn"
