export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for lib in librlrep_hip.so librlrep_hip_nou.so; do
  echo "== $lib"
  RLREP_LIB=$R/rlrep_amd/lib/$lib RLREP_STAMP=1 python3 tools/exp/chain_stamps.py 2>&1 | tail -n 5
  RLREP_LIB=$R/rlrep_amd/lib/$lib python3 bench.py --steps 2000 --warmup 200 --no-cpu --no-profile --quick 2>/dev/null | tail -n 1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("value", d["value"])'
done
