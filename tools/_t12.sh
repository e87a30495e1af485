export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
python3 -m pytest tests/test_gemm_engines.py -x -q -m gpu > gpurun_out/pytest_ge.log 2>&1; rc=$?; tail -n 15 gpurun_out/pytest_ge.log
[ $rc -ne 0 ] && exit $rc
python3 -m pytest tests -x -q -m gpu -k "ctrlsac or spedersac or diffsrsac or large_dims or golden or soak" > gpurun_out/pytest_d.log 2>&1; rc=$?; tail -n 8 gpurun_out/pytest_d.log
[ $rc -ne 0 ] && exit $rc
for w in spedersac_ant_f512_b1024 ctrlsac_halfcheetah_f2048_b256 diffsrsac_halfcheetah_b256 ctrlsac_halfcheetah_f256_b256; do
for rep in 1 2; do
for arm in "-" "RLREP_X3S_OFF=1"; do
  if [ "$arm" = "-" ]; then envs=""; else envs="$arm"; fi
  line=$(env $envs python3 bench.py --workload $w --steps 600 --warmup 60 --no-cpu --no-profile --quick 2>/dev/null | tail -n 1)
  echo "[ab] $w rep $rep arm [$arm]: $(echo $line | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done
done
done
