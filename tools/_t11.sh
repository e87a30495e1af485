export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
ENGINES=1,2 ONLY=diffsr python3 tools/bench_gemm.py 2>&1 | tee gpurun_out/gemm_engines_r04.txt
ENGINES=1,2 ONLY=square python3 tools/bench_gemm.py 2>&1 | tee -a gpurun_out/gemm_engines_r04.txt
RLREP_X3_DW_OLD=1 ENGINES=2 ONLY=diffsr python3 tools/bench_gemm.py 2>&1 | sed 's/^/[staging-transpose form] /' | tee -a gpurun_out/gemm_engines_r04.txt
