# round 6, first contact of the new data-parallel code: loopback tests, multi-process tests (timed per test), then the loopback cost measurement
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out
timeout -k 10 300 python3 -m pytest tests/test_loopback.py -x -q -v -m gpu --durations=20 > gpurun_out/pytest_loopback.log 2>&1; rc=$?; tail -n 30 gpurun_out/pytest_loopback.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 1500 python3 -m pytest tests/test_comm.py -q -m gpu --durations=40 > gpurun_out/pytest_comm.log 2>&1; rc=$?; tail -n 60 gpurun_out/pytest_comm.log
[ $rc -ne 0 ] && exit $rc
for w in 2; do
  timeout -k 10 300 python3 tools/exp/dp_loopback.py --workload vlsac_halfcheetah_f256_b256 --world $w --calls 400 > gpurun_out/loopback_vlsac_w$w.log 2>&1 || { tail -n 30 gpurun_out/loopback_vlsac_w$w.log; exit 1; }
  grep '^{' gpurun_out/loopback_vlsac_w$w.log
done
