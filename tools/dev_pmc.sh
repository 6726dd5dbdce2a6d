cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="7680 1920 3992 1 1 5 1 1"
python tools/dev_gemm_perf.py 7680 1920 3992 1 1 20 1 1
python tools/dev_gemm_perf.py 7680 1920 3992 1 1 20 1 0
python tools/dev_gemm_perf.py 7680 1920 3992 1 1 20 66 1
python tools/dev_gemm_perf.py 1920 7680 3992 1 1 20 1 1
python tools/dev_gemm_perf.py 1920 1920 3992 1 1 20 1 1
python tools/dev_gemm_perf.py 5760 1920 3992 1 1 20 1 1
python tools/dev_gemm_perf.py 3992 1920 5760 0 1 20 1 0
python tools/dev_gemm_perf.py 3992 7680 1920 0 1 20 1 0
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_tn_a -- python tools/dev_gemm_perf.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_LDS_UNALIGNED_STALL SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_tn_d -- python tools/dev_gemm_perf.py $ARGS > /dev/null 2>&1
