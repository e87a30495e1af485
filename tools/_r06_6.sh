# round 6: the 32 x 32 bf16x3 tile (tests, microbenchmark, workloads), kernel stats of one attached replica, bench --gpus 2 over gloo on one GPU
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gemm_engines.py -x -q -m gpu -k "32_by_32 or split_k" > gpurun_out/pytest_sub.log 2>&1; rc=$?; tail -n 8 gpurun_out/pytest_sub.log
[ $rc -ne 0 ] && exit $rc
python3 - <<'PY' | tee gpurun_out/x3q_micro.txt
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
timeout -k 10 600 python3 -m pytest tests/test_hip_parity.py tests/test_cross_terms.py tests/test_large_dims.py -x -q -m gpu > gpurun_out/pytest_sub2.log 2>&1; rc=$?; tail -n 5 gpurun_out/pytest_sub2.log
[ $rc -ne 0 ] && exit $rc
for w in ctrlsac_halfcheetah_f2048_b256 diffsrsac_halfcheetah_b256 spedersac_ant_f512_b1024; do
  for rep in 1 2; do
    for arm in "" "RLREP_DISABLE=x3q"; do
      env $arm python3 bench.py --workload $w --steps 400 --warmup 60 --no-cpu --quick --no-profile > gpurun_out/ab_tmp.log 2>&1 || { tail -n 20 gpurun_out/ab_tmp.log; exit 1; }
      echo "$w arm[$arm] $(tail -n 1 gpurun_out/ab_tmp.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d.get("launches_per_train"))')" | tee -a gpurun_out/ab_r06_x3q.txt
    done
  done
done
cd /tmp
for arm in alone alone_attached; do
  for w in vlsac_halfcheetah_f256_b256 ctrlsac_halfcheetah_f256_b256; do
    rm -rf $R/gpurun_out/prof_x
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_x -- python3 $R/tools/exp/dp_loopback.py --workload $w --arms $arm --calls 300 --warm 60 > $R/gpurun_out/prof_${arm}_$w.log 2>&1 || { tail -n 20 $R/gpurun_out/prof_${arm}_$w.log; exit 1; }
    f=$(ls $R/gpurun_out/prof_x/*/*kernel_stats.csv | head -1)
    [ -n "$f" ] && { echo "== $arm $w"; head -n 16 "$f" | cut -c1-170; cp "$f" $R/gpurun_out/stats_${arm}_$w.csv; }
    rm -rf $R/gpurun_out/prof_x
  done
done
cd $R
RLREP_DIST_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus 2 --steps 200 --warmup 50 --no-cpu --quick > gpurun_out/bench_gloo2.log 2>&1 || { tail -n 30 gpurun_out/bench_gloo2.log; exit 1; }
tail -n 1 gpurun_out/bench_gloo2.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["n_gpus"], d["replicas_identical"]); print(json.dumps(d["dp_forms"]))'
