# round 6: the three-rank headline case, the whole GPU suite (timed), the loopback cost measurement, the driver-form bench line
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out
timeout -k 10 400 python3 -m pytest tests/test_comm.py -x -q -m gpu -k "vlsac_hc-3" > gpurun_out/pytest_comm3.log 2>&1; rc=$?; tail -n 5 gpurun_out/pytest_comm3.log
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu --durations=25 > gpurun_out/pytest_gpu.log 2>&1; rc=$?; tail -n 40 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
for w in vlsac_halfcheetah_f256_b256 spedersac_ant_f512_b1024 ctrlsac_halfcheetah_f256_b256; do
  timeout -k 10 300 python3 tools/exp/dp_loopback.py --workload $w --world 2 --calls 300 > gpurun_out/loopback_$w.log 2>&1 || { tail -n 30 gpurun_out/loopback_$w.log; exit 1; }
  grep '^{' gpurun_out/loopback_$w.log
done
python3 bench.py --steps 20 --warmup 5 > gpurun_out/bench_driver.log 2>&1 || { tail -n 30 gpurun_out/bench_driver.log; exit 1; }
tail -n 1 gpurun_out/bench_driver.log | cut -c1-900
