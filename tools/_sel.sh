export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
python3 bench.py --workload ctrlsac_halfcheetah_f2048_b256 --steps 300 --warmup 50 --no-cpu 2>/dev/null | tail -n 1 > gpurun_out/ctrl_now.json
