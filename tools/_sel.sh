export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/sel_tests.log 2>&1 || { tail -n 30 gpurun_out/sel_tests.log; exit 1; }
tail -n 2 gpurun_out/sel_tests.log
for i in 1 2 3; do python3 bench.py --steps 20 --warmup 5 --no-cpu --no-profile 2>/dev/null | tail -n 1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("[driver form]", d["value"], d.get("value_median_500"), d.get("main_loop_iterations_per_sec"))'; done
timeout -k 10 300 python3 tools/exp/host_bound.py vlsac_halfcheetah_f256_b256 2>&1 | sed -n 2,4p
