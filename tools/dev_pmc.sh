cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "7680 1920 3992 1 1 5 3 1" "3992 7680 1920 0 0 5 3 0"; do
tag=$(echo $cfg | tr ' ' '_')
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmcx_a_$tag -- python tools/dev_gemm_perf.py $cfg > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_LDS_UNALIGNED_STALL SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmcx_d_$tag -- python tools/dev_gemm_perf.py $cfg > /dev/null 2>&1
done
