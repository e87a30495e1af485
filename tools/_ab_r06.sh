# The alternated A/B runs docs/history/r06.md quotes (run on the GPU box: gpurun -- 'bash tools/_ab_r06.sh [part]'); one box per part.
#   head  : vlsac headline: default / RLREP_DISABLE=dw_xcd / RLREP_ENABLE=nc_u_nt
#   split : workloads with split-K stages: default / RLREP_ENABLE=fin_inline / RLREP_DISABLE=x3q
#   k12   : ctrlsac config 3: default / RLREP_ENABLE=fuse_infonce (the score matrix inside the InfoNCE launch)
#   micro : the 32 x 32 bf16x3 tile against the 64-wide tile + finisher through rlrep_gemm
#   dp    : tools/exp/dp_loopback.py: protocol-only cost and two replicas on one chip (-> gpurun_out/r06_dp_loopback.txt)
#   soak  : long attached runs, replicas_identical
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
PART=${1:-all}
line() { tail -n 1 gpurun_out/ab_tmp.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d.get("launches_per_train"))'; }
if [ $PART = head ] || [ $PART = all ]; then
  for rep in 1 2; do for arm in "" "RLREP_DISABLE=dw_xcd" "RLREP_ENABLE=nc_u_nt"; do
    env $arm python3 bench.py --steps 2000 --warmup 300 --no-cpu --quick --no-profile > gpurun_out/ab_tmp.log 2>&1 || exit 1
    echo "vlsac arm[$arm] $(line)"
  done; done
fi
if [ $PART = split ] || [ $PART = all ]; then
  for w in ctrlsac_halfcheetah_f2048_b256 spedersac_ant_f512_b1024 diffsrsac_halfcheetah_b256; do for rep in 1 2; do for arm in "" "RLREP_ENABLE=fin_inline" "RLREP_DISABLE=x3q"; do
    env $arm python3 bench.py --workload $w --steps 400 --warmup 60 --no-cpu --quick --no-profile > gpurun_out/ab_tmp.log 2>&1 || exit 1
    echo "$w arm[$arm] $(line)"
  done; done; done
fi
if [ $PART = k12 ] || [ $PART = all ]; then
  for rep in 1 2 3; do for arm in "" "RLREP_ENABLE=fuse_infonce"; do
    env $arm python3 bench.py --workload ctrlsac_halfcheetah_f256_b256 --steps 2000 --warmup 300 --no-cpu --quick --no-profile > gpurun_out/ab_tmp.log 2>&1 || exit 1
    echo "ctrlsac F=256 arm[$arm] $(line)"
  done; done
fi
if [ $PART = micro ] || [ $PART = all ]; then
  python3 - <<'PY'
import sys
sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import bench_gemm as bg
for name, mode, R, Cn, K in [('phi.l2 fwd', 'fwd', 256, 1024, 1024), ('phi.l3 fwd', 'fwd', 256, 2048, 1024), ('critic l1|l4 fwd', 'fwd', 256, 2048, 2048),
                             ('critic l1|l4 dx', 'dx', 256, 2048, 2048), ('phi.l3 dx', 'dx', 256, 1024, 2048), ('phi.l2 dx', 'dx', 256, 1024, 1024),
                             ('speder phi fwd', 'fwd', 2048, 512, 512), ('speder critic fwd', 'fwd', 1024, 512, 512)]:
    a = bg.run(2, mode, R, Cn, K, 200, bt=32, splits=0)
    b = bg.run(2, mode, R, Cn, K, 200, bt=64, splits=0)
    gf = 2.0 * R * Cn * K / 1e9
    print(f'{name:20s} {mode} {R}x{Cn}x{K}: 32x32 tile {a:6.1f} us ({gf / a * 1e3:6.1f} TF)   64-wide tile + finisher {b:6.1f} us ({gf / b * 1e3:6.1f} TF)', flush=True)
PY
fi
if [ $PART = dp ] || [ $PART = all ]; then
  : > gpurun_out/r06_dp_loopback.txt
  for w in vlsac_halfcheetah_f256_b256 spedersac_ant_f512_b1024 ctrlsac_halfcheetah_f256_b256 sac_halfcheetah_b256; do
    python3 tools/exp/dp_loopback.py --workload $w --arms alone,alone_attached --calls 400 2>/dev/null | grep '^{' | tee -a gpurun_out/r06_dp_loopback.txt
  done
  for w in vlsac_halfcheetah_f256_b256 spedersac_ant_f512_b1024 ctrlsac_halfcheetah_f256_b256; do
    python3 tools/exp/dp_loopback.py --workload $w --world 2 --calls 300 2>/dev/null | grep '^{' | tee -a gpurun_out/r06_dp_loopback.txt
  done
fi
if [ $PART = soak ] || [ $PART = all ]; then
  : > gpurun_out/r06_dp_soak.txt
  python3 tools/exp/dp_loopback.py --workload vlsac_halfcheetah_f256_b256 --arms attached --calls 20000 --warm 200 2>/dev/null | grep '^{' | tee -a gpurun_out/r06_dp_soak.txt
  python3 tools/exp/dp_loopback.py --workload spedersac_ant_f512_b1024 --arms attached --calls 5000 --warm 100 2>/dev/null | grep '^{' | tee -a gpurun_out/r06_dp_soak.txt
  python3 tools/exp/dp_loopback.py --workload ctrlsac_halfcheetah_f256_b256 --arms attached --calls 10000 --warm 100 2>/dev/null | grep '^{' | tee -a gpurun_out/r06_dp_soak.txt
  python3 tools/exp/dp_loopback.py --workload sac_halfcheetah_b256 --arms attached --calls 20000 --warm 100 2>/dev/null | grep '^{' | tee -a gpurun_out/r06_dp_soak.txt
fi
