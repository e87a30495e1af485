export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for arm in "-" "RLREP_X3S_OFF=1" "RLREP_PIPELINE=0" "RLREP_PIPELINE=0 RLREP_X3S_OFF=1"; do
  if [ "$arm" = "-" ]; then envs=""; else envs="$arm"; fi
  env $envs python3 bench.py --workload spedersac_ant_f512_b1024 --steps 600 --warmup 60 --no-cpu 2>/dev/null | tail -n 1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d.get("chains"), [ (f["family"], f["launches_per_train"], f["us_per_train"]) for f in d["kernel_families"]])' "$arm"
done
