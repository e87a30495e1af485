export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT
export ONLY="${ONLY:-square}" ENGINES="${ENGINES:-1,2}"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_gemm -- python3 $R/tools/bench_gemm.py > $R/gpurun_out/pmc_gemm.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_gemm2 -- python3 $R/tools/bench_gemm.py > $R/gpurun_out/pmc_gemm2.log 2>&1
echo done
