# round 6: in-launch split-K finish (tests + A/B on the workloads that split), the protocol-only cost of the attached form
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_large_dims.py tests/test_default_mode.py tests/test_switches.py -x -q -m gpu > gpurun_out/pytest_sub.log 2>&1; rc=$?; tail -n 8 gpurun_out/pytest_sub.log
[ $rc -ne 0 ] && exit $rc
for w in ctrlsac_halfcheetah_f2048_b256 spedersac_ant_f512_b1024 diffsrsac_halfcheetah_b256; do
  for rep in 1 2; do
    for arm in "" "RLREP_DISABLE=fin_inline"; do
      env $arm python3 bench.py --workload $w --steps 400 --warmup 60 --no-cpu --quick --no-profile > gpurun_out/ab_tmp.log 2>&1 || { tail -n 20 gpurun_out/ab_tmp.log; exit 1; }
      echo "$w arm[$arm] $(tail -n 1 gpurun_out/ab_tmp.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d.get("launches_per_train"))')" | tee -a gpurun_out/ab_r06_fin.txt
    done
  done
done
for w in vlsac_halfcheetah_f256_b256 spedersac_ant_f512_b1024 ctrlsac_halfcheetah_f256_b256 sac_halfcheetah_b256; do
  timeout -k 10 300 python3 tools/exp/dp_loopback.py --workload $w --arms alone,alone_attached --calls 400 > gpurun_out/loopback_alone_$w.log 2>&1 || { tail -n 30 gpurun_out/loopback_alone_$w.log; exit 1; }
  grep '^{' gpurun_out/loopback_alone_$w.log | tee -a gpurun_out/loopback_alone_all.txt
done
