export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for w in vlsac_halfcheetah_f256_b256 sac_halfcheetah_b256 sac_pendulum_b64 ctrlsac_halfcheetah_f256_b256 ctrlsac_halfcheetah_f2048_b256 spedersac_ant_f512_b1024 diffsrsac_halfcheetah_b256; do
  python3 bench.py --workload $w --no-cpu --no-profile 2>gpurun_out/t29_err.log | tail -n 1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("[bench]", d["config"]["workload"], d["value"], d.get("value_median_500", d.get("value_median_repeats")), d.get("main_loop_iterations_per_sec"), d["steps"])' || tail -n 3 gpurun_out/t29_err.log
done
timeout -k 10 300 python3 bench.py --workload diffsrsac_humanoid_b2048 --steps 20 --warmup 3 --no-cpu --no-profile 2>gpurun_out/t29_err.log | tail -n 1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("[bench]", d["config"]["workload"], d["value"], d.get("main_loop_iterations_per_sec"), d["steps"])'
