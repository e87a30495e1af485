# A/B of two workloads between the round-2 head (a git worktree built into _ab/r02 in the build container) and the current tree, ALTERNATED
# on one box (ctrlsac F=2048 767 -> 738, Humanoid 46.2 -> 44.5 were measured on different boxes).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/ab_r02.jsonl; : > $OUT
for rep in 1 2 3; do
  for tree in r02 cur; do
    if [ $tree = r02 ]; then D=$R/_ab/r02; else D=$R; fi
    cd $D
    for spec in "ctrlsac_halfcheetah_f2048_b256 600 60" "diffsrsac_humanoid_b2048 20 3"; do
      set -- $spec
      line=$(python3 bench.py --workload $1 --steps $2 --warmup $3 --no-cpu --no-profile 2>/dev/null | tail -n 1)
      echo "{\"tree\": \"$tree\", \"rep\": $rep, \"line\": $line}" >> $OUT
      echo "[ab] $tree rep $rep $1: $(echo $line | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d.get("value_median_500", d.get("value_median_repeats")))')"
    done
  done
done
