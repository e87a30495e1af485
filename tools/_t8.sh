export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for arm in "-" "RLREP_NO_CHAIN_NEXT=1"; do
  if [ "$arm" = "-" ]; then envs=""; else envs="$arm"; fi
  env $envs python3 bench.py --steps 1000 --warmup 100 --no-cpu 2>/dev/null | tail -n 1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["chains"], [ (f["family"], f["launches_per_train"], f["us_per_train"]) for f in d["kernel_families"]])' "$arm"
done
