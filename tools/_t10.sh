export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
python3 -m pytest tests/test_gemm_engines.py -x -q -m gpu > gpurun_out/pytest_ge.log 2>&1; rc=$?; tail -n 15 gpurun_out/pytest_ge.log
[ $rc -ne 0 ] && exit $rc
python3 -m pytest tests -x -q -m gpu -k "diffsrsac or humanoid or large_dims" > gpurun_out/pytest_d.log 2>&1; rc=$?; tail -n 8 gpurun_out/pytest_d.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do
for arm in "-" "RLREP_X3_DW_FP32=1" "RLREP_X3_DW_OLD=1"; do
  if [ "$arm" = "-" ]; then envs=""; else envs="$arm"; fi
  line=$(env $envs python3 bench.py --workload diffsrsac_humanoid_b2048 --steps 20 --warmup 3 --no-cpu --no-profile --quick 2>/dev/null | tail -n 1)
  echo "[ab] rep $rep arm [$arm]: $(echo $line | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done
done
