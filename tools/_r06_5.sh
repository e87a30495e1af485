# round 6: after the slot-sum launches / plain epoch load: tests, protocol-only cost, kernel stats of the attached replica, bench --gpus 2 over gloo on one GPU
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_large_dims.py tests/test_default_mode.py tests/test_loopback.py tests/test_comm.py -x -q -m gpu > gpurun_out/pytest_sub.log 2>&1; rc=$?; tail -n 8 gpurun_out/pytest_sub.log
[ $rc -ne 0 ] && exit $rc
rm -f gpurun_out/loopback_alone_all.txt
for w in vlsac_halfcheetah_f256_b256 spedersac_ant_f512_b1024 ctrlsac_halfcheetah_f256_b256 sac_halfcheetah_b256; do
  timeout -k 10 300 python3 tools/exp/dp_loopback.py --workload $w --arms alone,alone_attached --calls 400 > gpurun_out/loopback_alone_$w.log 2>&1 || { tail -n 30 gpurun_out/loopback_alone_$w.log; exit 1; }
  grep '^{' gpurun_out/loopback_alone_$w.log | tee -a gpurun_out/loopback_alone_all.txt
done
cd /tmp
for arm in alone alone_attached; do
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$arm -o x -- python3 $R/tools/exp/dp_loopback.py --workload vlsac_halfcheetah_f256_b256 --arms $arm --calls 300 --warm 60 > $R/gpurun_out/prof_$arm.log 2>&1 || { tail -n 20 $R/gpurun_out/prof_$arm.log; exit 1; }
  f=$(find $R/gpurun_out/prof_$arm -name '*kernel_stats.csv' | head -n 1); echo "== $arm"; head -n 14 $f | cut -c1-150
done
cd $R
RLREP_DIST_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus 2 --steps 200 --warmup 50 --no-cpu --quick > gpurun_out/bench_gloo2.log 2>&1 || { tail -n 30 gpurun_out/bench_gloo2.log; exit 1; }
tail -n 1 gpurun_out/bench_gloo2.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["n_gpus"], d["replicas_identical"]); print(json.dumps(d["dp_forms"]))'
