# re-collect ONE workload's bench line into gpurun_out/profiles_r05/r05_bench_all.jsonl (replacing its line): bash tools/_recollect_one.sh <workload> [steps]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
W=$1; ST=${2:-1000}
OUT=$R/gpurun_out/profiles_r05; mkdir -p $OUT
python3 bench.py --workload $W --steps $ST --warmup 100 > gpurun_out/bench_$W.log 2>&1 || { tail -n 5 gpurun_out/bench_$W.log; exit 1; }
python3 - "$W" <<'PY'
import json, sys
w = sys.argv[1]
new = open(f'gpurun_out/bench_{w}.log').read().strip().splitlines()[-1]
json.loads(new)
lines = [l.rstrip('\n') for l in open('profiles/r05_bench_all.jsonl')]
out = [new if json.loads(l)['config']['workload'] == w else l for l in lines]
open('gpurun_out/profiles_r05/r05_bench_all.jsonl', 'w').write('\n'.join(out) + '\n')
print('[recollect]', w, json.loads(new)['value'])
PY
