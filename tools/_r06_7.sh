# round 6: exchange primitives without cache maintenance: tests + protocol-only cost + kernel stats + two replicas on one chip
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_loopback.py tests/test_comm.py tests/test_gemm_engines.py -x -q -m gpu > gpurun_out/pytest_sub.log 2>&1; rc=$?; tail -n 8 gpurun_out/pytest_sub.log
[ $rc -ne 0 ] && exit $rc
rm -f gpurun_out/loopback_alone_all.txt gpurun_out/loopback_pair_all.txt
for w in vlsac_halfcheetah_f256_b256 spedersac_ant_f512_b1024 ctrlsac_halfcheetah_f256_b256 sac_halfcheetah_b256; do
  timeout -k 10 300 python3 tools/exp/dp_loopback.py --workload $w --arms alone,alone_attached --calls 400 > gpurun_out/loopback_alone_$w.log 2>&1 || { tail -n 30 gpurun_out/loopback_alone_$w.log; exit 1; }
  grep '^{' gpurun_out/loopback_alone_$w.log | tee -a gpurun_out/loopback_alone_all.txt
done
for w in vlsac_halfcheetah_f256_b256 spedersac_ant_f512_b1024 ctrlsac_halfcheetah_f256_b256; do
  timeout -k 10 300 python3 tools/exp/dp_loopback.py --workload $w --world 2 --calls 300 > gpurun_out/loopback_$w.log 2>&1 || { tail -n 30 gpurun_out/loopback_$w.log; exit 1; }
  grep '^{' gpurun_out/loopback_$w.log | tee -a gpurun_out/loopback_pair_all.txt
done
cd /tmp
for arm in alone_attached; do
  for w in vlsac_halfcheetah_f256_b256 ctrlsac_halfcheetah_f256_b256; do
    rm -rf $R/gpurun_out/prof_x
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_x -- python3 $R/tools/exp/dp_loopback.py --workload $w --arms $arm --calls 300 --warm 60 > $R/gpurun_out/prof_${arm}_$w.log 2>&1 || { tail -n 20 $R/gpurun_out/prof_${arm}_$w.log; exit 1; }
    f=$(ls $R/gpurun_out/prof_x/*/*kernel_stats.csv | head -1)
    [ -n "$f" ] && { echo "== $arm $w"; grep -E "adam|comm_" "$f" | cut -c1-60,200-400; cp "$f" $R/gpurun_out/stats_${arm}_$w.csv; }
    rm -rf $R/gpurun_out/prof_x
  done
done
