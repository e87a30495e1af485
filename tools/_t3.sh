export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for arm in "-" "RLREP_NO_CHAIN_NEXT=1" "RLREP_NO_FOLD_NCDW=1" "RLREP_NO_CHAIN_NEXT=1 RLREP_NO_FOLD_NCDW=1" "RLREP_MANAGED_IMAGES=1"; do
  if [ "$arm" = "-" ]; then envs=""; else envs="$arm"; fi
  echo "== arm [$arm]"
  env $envs RLREP_STAMP=1 python3 tools/exp/chain_stamps.py 2>&1 | tail -n 5
done
