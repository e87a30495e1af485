# Collect everything profiles/r03_* is built from (run on the GPU box: gpurun -- 'bash tools/_collect_r03.sh')
export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03b -- python3 $R/bench.py --steps 300 --warmup 30 --no-cpu --no-profile > $R/gpurun_out/prof_r03b.log 2>&1 || exit 1
echo "[collect] kernel trace done"
bash $R/tools/_pmc.sh > $R/gpurun_out/pmc.log 2>&1 || exit 1
echo "[collect] pmc done"
cd /tmp
python3 $R/bench.py > $R/gpurun_out/bench_r03b.log 2>&1 || exit 1
tail -n 1 $R/gpurun_out/bench_r03b.log > $R/gpurun_out/bench_r03b.json
python3 $R/bench.py --steps 20 --warmup 5 > $R/gpurun_out/bench_r03b_driver.log 2>&1 || exit 1
echo "[collect] bench done"
for w in ctrlsac_halfcheetah_f2048_b256 ctrlsac_halfcheetah_f256_b256 spedersac_ant_f512_b1024 sac_halfcheetah_b256 sac_pendulum_b64 diffsrsac_halfcheetah_b256; do python3 $R/bench.py --workload $w --steps 1000 --warmup 100 --no-cpu > $R/gpurun_out/bench_$w.log 2>&1 || exit 1; echo "[collect] $w"; done
python3 $R/bench.py --workload diffsrsac_humanoid_b2048 --steps 20 --warmup 3 --no-cpu > $R/gpurun_out/bench_diffsrsac_humanoid_b2048.log 2>&1 || exit 1
# in-kernel timelines of the tile engine (instrumented builds, if present)
if [ -f $R/rlrep_amd/lib/librlrep_hip_tim.so ]; then RLREP_LIB=$R/rlrep_amd/lib/librlrep_hip_tim.so RLREP_PIPELINE=0 python3 $R/tools/exp/gemm_timeline.py > $R/gpurun_out/r03_gemm_timeline.log 2>&1; fi
# the one-rank RCCL rehearsals of the data-parallel forms
RLREP_FORCE_DP=1 python3 $R/bench.py --steps 2000 --warmup 200 --no-cpu --no-profile > $R/gpurun_out/r03_dp_seq_captured.log 2>&1
RLREP_FORCE_DP=1 RLREP_DP_CAPTURE=0 python3 $R/bench.py --steps 2000 --warmup 200 --no-cpu --no-profile > $R/gpurun_out/r03_dp_seq_segments.log 2>&1
RLREP_FORCE_DP=1 RLREP_PIPELINE_DP=1 python3 $R/bench.py --steps 2000 --warmup 200 --no-cpu --no-profile > $R/gpurun_out/r03_dp_pipe_captured.log 2>&1
# summarise ON THE BOX (the raw rocprofv3 trees are > 64 MiB: only the summaries travel back), then drop the raw trees
RLREP_PROFILES_OUT=$R/gpurun_out/profiles_r03 python3 $R/tools/summarize_profiles.py r03 > $R/gpurun_out/profiles_r03_summary.txt 2>&1
rm -rf $R/gpurun_out/prof_r03b $R/gpurun_out/pmc_sq $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
echo collected
