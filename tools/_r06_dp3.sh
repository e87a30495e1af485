# round 6: loopback tests inside the whole suite's process, the loopback cost measurement, A/B of the two new single-GPU switches, driver-form line
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_hip_parity.py tests/test_loopback.py tests/test_default_mode.py -x -q -m gpu > gpurun_out/pytest_sub.log 2>&1; rc=$?; tail -n 8 gpurun_out/pytest_sub.log
[ $rc -ne 0 ] && exit $rc
for w in vlsac_halfcheetah_f256_b256 spedersac_ant_f512_b1024 ctrlsac_halfcheetah_f256_b256; do
  timeout -k 10 300 python3 tools/exp/dp_loopback.py --workload $w --world 2 --calls 300 > gpurun_out/loopback_$w.log 2>&1 || { tail -n 30 gpurun_out/loopback_$w.log; exit 1; }
  grep '^{' gpurun_out/loopback_$w.log
done
# A/B, alternated: default (dw tiles as XCD runs) / RLREP_DISABLE=dw_xcd / RLREP_ENABLE=nc_u_nt
for rep in 1 2; do
  for arm in "" "RLREP_DISABLE=dw_xcd" "RLREP_ENABLE=nc_u_nt"; do
    env $arm python3 bench.py --steps 2000 --warmup 300 --no-cpu --quick --no-profile > gpurun_out/ab_tmp.log 2>&1 || { tail -n 20 gpurun_out/ab_tmp.log; exit 1; }
    echo "arm[$arm] $(tail -n 1 gpurun_out/ab_tmp.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')" | tee -a gpurun_out/ab_r06_1.txt
  done
done
python3 bench.py --steps 20 --warmup 5 > gpurun_out/bench_driver.log 2>&1 || { tail -n 30 gpurun_out/bench_driver.log; exit 1; }
tail -n 1 gpurun_out/bench_driver.log | cut -c1-700
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_driver.log').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('value','value_median_500','main_loop_iterations_per_sec','roofline')})
print(d.get('stage_times_us',{}).get('gemm16'))
print(d.get('chains'))
PY
