export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT
for w in ${WL:-ctrlsac_halfcheetah_f2048_b256:60:10 spedersac_ant_f512_b1024:60:10 diffsrsac_humanoid_b2048:6:2}; do
  IFS=: read name steps warm <<< "$w"
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$name -- python3 $R/bench.py --workload $name --steps $steps --warmup $warm --no-cpu > $R/gpurun_out/prof_$name.log 2>&1 || exit 1
  python3 $R/bench.py --workload $name --steps $((steps*3)) --warmup $warm --no-cpu > $R/gpurun_out/bench_$name.log 2>&1 || exit 1
done
