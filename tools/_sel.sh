export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 900 python3 -m pytest tests/test_launcher.py tests/test_cross_terms.py tests/test_hip_parity.py -m gpu -x -q > gpurun_out/sel_tests.log 2>&1 || { tail -n 30 gpurun_out/sel_tests.log; exit 1; }
tail -n 2 gpurun_out/sel_tests.log
timeout -k 10 300 python3 tools/exp/host_loop.py > gpurun_out/host_loop2.txt 2>&1; head -n 4 gpurun_out/host_loop2.txt
python3 bench.py --steps 2000 --warmup 200 --no-cpu --no-profile 2>/dev/null | tail -n 1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("[bench]", d["value"], d.get("value_median_500"), d.get("value_with_replay_add_per_call"), d.get("main_loop_iterations_per_sec"))'
