# GPU suite + bench lines (driver form + two other workloads) -- a quick validation call
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
python3 -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; rc=$?; tail -n 15 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
python3 bench.py --steps 20 --warmup 5 > gpurun_out/bench_driver.log 2>&1 || { tail -n 30 gpurun_out/bench_driver.log; exit 1; }
tail -n 1 gpurun_out/bench_driver.log | cut -c1-1500
for w in ${CHECK_WORKLOADS:-spedersac_ant_f512_b1024 sac_halfcheetah_b256}; do
  python3 bench.py --workload $w --steps 300 --warmup 30 > gpurun_out/bench_$w.log 2>&1 || { tail -n 30 gpurun_out/bench_$w.log; exit 1; }
  tail -n 1 gpurun_out/bench_$w.log | cut -c1-1200
done
