# Collect everything profiles/r01_* is built from (run on the GPU box: gpurun -- 'bash tools/_collect_r01.sh')
export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01b -- python3 $R/bench.py --steps 300 --warmup 30 --no-cpu > $R/gpurun_out/prof_r01b.log 2>&1 || exit 1
bash $R/tools/_pmc.sh > $R/gpurun_out/pmc.log 2>&1 || exit 1
cd /tmp
python3 $R/bench.py > $R/gpurun_out/bench_r01b.log 2>&1 || exit 1
tail -n 1 $R/gpurun_out/bench_r01b.log > $R/gpurun_out/bench_r01b.json
WL="ctrlsac_halfcheetah_f2048_b256:60:10 spedersac_ant_f512_b1024:60:10 diffsrsac_humanoid_b2048:6:2" bash $R/tools/_prof_big.sh || exit 1
for w in ctrlsac_halfcheetah_f256_b256 sac_halfcheetah_b256 sac_pendulum_b64 diffsrsac_halfcheetah_b256; do python3 $R/bench.py --workload $w --steps 2000 --warmup 200 --no-cpu > $R/gpurun_out/bench_$w.log 2>&1 || exit 1; done
ONLY="" ENGINES=1,2 bash $R/tools/_pmc_gemm.sh
echo collected
