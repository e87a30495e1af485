export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
L=$R/rlrep_amd/lib
for lib in librlrep_hip_bd0bc5b.so librlrep_hip_20dcc8a.so; do
  RLREP_LIB=$L/$lib python3 bench.py --workload ctrlsac_halfcheetah_f2048_b256 --steps 300 --warmup 50 --no-cpu 2>/dev/null | tail -n 1 > gpurun_out/t33_$lib.json
done
