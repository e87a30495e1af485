export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=$R/gpurun_out/profiles_r04; mkdir -p $OUT
timeout -k 10 500 python3 bench.py --workload diffsrsac_humanoid_b2048 --steps 20 --warmup 3 > gpurun_out/bench_diffsrsac_humanoid_b2048.log 2>&1 || { tail -n 5 gpurun_out/bench_diffsrsac_humanoid_b2048.log; exit 1; }
tail -n 1 gpurun_out/bench_diffsrsac_humanoid_b2048.log > $OUT/r04_bench_humanoid.json
tail -n 1 gpurun_out/bench_diffsrsac_humanoid_b2048.log | cut -c1-1200
