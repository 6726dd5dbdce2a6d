#!/bin/bash
# MFMA-busy per launch shape of the three training workloads (automatic kernel choice): one --pmc pass, no tracing.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HIP_FORCE_DEV_KERNARG=1
rm -rf gpurun_out/r05_gemm_pmc
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r05_gemm_pmc -- python3 tools/r05_gemm_table.py --kernels 0 --iters 3 > gpurun_out/r05_gemm_pmc_table.txt 2>&1
f=$(find gpurun_out/r05_gemm_pmc -name "*counter_collection.csv" | head -1)
python3 tools/r05_gemm_pmc.py $f gpurun_out/r05_gemm_pmc_table.txt > gpurun_out/r05_gemm_pmc_summary.txt 2>&1
cat gpurun_out/r05_gemm_pmc_summary.txt
find gpurun_out/r05_gemm_pmc -name "*.csv" -size +8M -delete; find gpurun_out/r05_gemm_pmc -name "*.db" -delete
