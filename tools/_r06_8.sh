# round 6: the 32 x 32 tile with three register sets: tests, microbenchmark, alternated A/B on the workloads it serves; headline sanity
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gemm_engines.py tests/test_hip_parity.py tests/test_cross_terms.py tests/test_default_mode.py -x -q -m gpu > gpurun_out/pytest_sub.log 2>&1; rc=$?; tail -n 6 gpurun_out/pytest_sub.log
[ $rc -ne 0 ] && exit $rc
python3 - <<'PY' | tee gpurun_out/x3q_micro3.txt
import os, sys
sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import bench_gemm as bg
for name, mode, R, Cn, K in [('phi.l2 fwd', 'fwd', 256, 1024, 1024), ('phi.l3 fwd', 'fwd', 256, 2048, 1024), ('critic l1|l4 fwd', 'fwd', 256, 2048, 2048),
                             ('critic l1|l4 dx', 'dx', 256, 2048, 2048), ('phi.l3 dx', 'dx', 256, 1024, 2048), ('phi.l2 dx', 'dx', 256, 1024, 1024)]:
    a = bg.run(2, mode, R, Cn, K, 200, bt=32, splits=0)
    b = bg.run(2, mode, R, Cn, K, 200, bt=64, splits=0)
    gf = 2.0 * R * Cn * K / 1e9
    print(f'{name:20s} {mode} {R}x{Cn}x{K}: 32x32 tile {a:6.1f} us ({gf / a * 1e3:6.1f} TF)   64-wide tile + finisher {b:6.1f} us ({gf / b * 1e3:6.1f} TF)', flush=True)
PY
for w in ctrlsac_halfcheetah_f2048_b256 diffsrsac_halfcheetah_b256; do
  for rep in 1 2 3; do
    for arm in "" "RLREP_DISABLE=x3q"; do
      env $arm python3 bench.py --workload $w --steps 400 --warmup 60 --no-cpu --quick --no-profile > gpurun_out/ab_tmp.log 2>&1 || { tail -n 20 gpurun_out/ab_tmp.log; exit 1; }
      echo "$w arm[$arm] $(tail -n 1 gpurun_out/ab_tmp.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d.get("launches_per_train"))')" | tee -a gpurun_out/ab_r06_x3q3.txt
    done
  done
done
python3 bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/bench_driver.log 2>&1 || { tail -n 30 gpurun_out/bench_driver.log; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_driver.log').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('value','value_median_500','main_loop_iterations_per_sec')}, d['roofline']['frac'], d['roofline']['traffic_detail'].get('coverage'), d['roofline']['traffic_detail'].get('launches_per_train_timed_form'))
PY
