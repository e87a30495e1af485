export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 600 python3 -m pytest tests/test_hip_parity.py tests/test_default_mode.py -m gpu -x -q > gpurun_out/t28_tests.log 2>&1 || { tail -n 30 gpurun_out/t28_tests.log; exit 1; }
tail -n 2 gpurun_out/t28_tests.log
for i in 1 2 3; do python3 bench.py --steps 3000 --warmup 200 --no-cpu --no-profile 2>/dev/null | tail -n 1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("[bench]", d["value"], d.get("value_median_500"), d.get("main_loop_iterations_per_sec"))'; done
RLREP_LIB=$R/rlrep_amd/lib/librlrep_hip_tim.so timeout -k 10 200 python3 tools/exp/gemm_timeline.py > gpurun_out/gemm_timeline4.txt 2>&1
