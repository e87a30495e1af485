# A/B of environment switches on ONE box, alternated: bash tools/_ab_env.sh "<workload>" "<steps>" "NAME1=VAL1 NAME2=VAL2" "NAMEX=VALX" ...
# each remaining argument is one arm (a space-separated list of VAR=VALUE, or "-" for the default)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
W=$1; ST=$2; shift 2
for rep in 1 2 3; do
  for arm in "$@"; do
    if [ "$arm" = "-" ]; then envs=""; else envs="$arm"; fi
    line=$(env $envs python3 bench.py --workload $W --steps $ST --warmup 200 --no-cpu --no-profile 2>gpurun_out/ab_err.log | tail -n 1)
    echo "[ab] rep $rep arm [$arm]: $(echo $line | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d.get("value_median_500", d.get("value_median_repeats")), d.get("main_loop_iterations_per_sec"))' 2>/dev/null || tail -n 3 gpurun_out/ab_err.log)"
  done
done
