export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
L=$R/rlrep_amd/lib
for lib in librlrep_hip_bd0bc5b.so librlrep_hip.so librlrep_hip_bd0bc5b.so librlrep_hip.so; do
  echo "== $lib"; RLREP_LIB=$L/$lib python3 tools/bench_gemm.py 2>/dev/null | grep -E "ctrlsac|spedersac phi" | grep -E "engine 2|x3|bf16" | head -12
done
