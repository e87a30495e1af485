export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
X3_MIN_FLOP=1e8 BT=64 ENGINES=1,2 ONLY=ctrlsac python3 tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids
X3_MIN_FLOP=1e8 BT=64 ENGINES=1,2 ONLY=spedersac python3 tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids
