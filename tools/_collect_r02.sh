# Collect everything profiles/r02_* is built from (run on the GPU box: gpurun -- 'bash tools/_collect_r02.sh')
export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02b -- python3 $R/bench.py --steps 300 --warmup 30 --no-cpu --no-profile > $R/gpurun_out/prof_r02b.log 2>&1 || exit 1
echo "[collect] kernel trace done"
bash $R/tools/_pmc.sh > $R/gpurun_out/pmc.log 2>&1 || exit 1
echo "[collect] pmc done"
cd /tmp
python3 $R/bench.py > $R/gpurun_out/bench_r02b.log 2>&1 || exit 1
tail -n 1 $R/gpurun_out/bench_r02b.log > $R/gpurun_out/bench_r02b.json
echo "[collect] bench done"
for w in ctrlsac_halfcheetah_f2048_b256 ctrlsac_halfcheetah_f256_b256 spedersac_ant_f512_b1024 sac_halfcheetah_b256 sac_pendulum_b64 diffsrsac_halfcheetah_b256; do python3 $R/bench.py --workload $w --steps 1000 --warmup 100 --no-cpu > $R/gpurun_out/bench_$w.log 2>&1 || exit 1; echo "[collect] $w"; done
python3 $R/bench.py --workload diffsrsac_humanoid_b2048 --steps 20 --warmup 3 --no-cpu > $R/gpurun_out/bench_diffsrsac_humanoid_b2048.log 2>&1 || exit 1
# the opt-in row-program form of the feature step: per-op timeline + the same bench line
RLREP_ROWPROG=1 python3 $R/tools/exp/rp_timeline.py > $R/gpurun_out/r02_rowprog_timeline.log 2>&1 || exit 1
RLREP_ROWPROG=1 python3 $R/bench.py --steps 1000 --warmup 100 --no-cpu > $R/gpurun_out/r02_rowprog_bench.log 2>&1 || exit 1
echo collected
