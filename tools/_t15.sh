export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for sp in 0 1 2 4 8 16; do
  echo "== splits $sp"
  SPLITS=$sp X3_MIN_FLOP=1e8 BT=64 ENGINES=2 ONLY=ctrlsac python3 tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids
  SPLITS=$sp X3_MIN_FLOP=1e8 BT=64 ENGINES=2 ONLY="spedersac phi" python3 tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids
done
