export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu --no-graph --no-profile > $R/gpurun_out/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu --no-graph --no-profile > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu --no-graph --no-profile > $R/gpurun_out/pmc_write.log 2>&1
ls $R/gpurun_out/pmc_sq/*/ | head; tail -2 $R/gpurun_out/pmc_sq.log | cut -c1-200
