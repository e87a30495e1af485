export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT
for w in spedersac_ant_f512_b1024 diffsrsac_halfcheetah_b256 ctrlsac_halfcheetah_f256_b256; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_x -- python3 $R/bench.py --workload $w --steps 200 --warmup 20 --no-cpu --no-profile > $R/gpurun_out/prof_x.log 2>&1
  f=$(ls $R/gpurun_out/prof_x/*/*kernel_stats.csv | head -1)
  echo "== $w"; tail -1 $R/gpurun_out/prof_x.log | cut -c60-130
  python3 - "$f" <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:14]:
    n=re.sub(r'\(.*$','',r['Name']).replace('void ','')[:60]
    print(f"{n:62s} calls {int(r['Calls']):7d} avg {float(r['AverageNs'])/1e3:7.2f} us  {float(r['Percentage']):5.1f} %")
PY
  rm -rf $R/gpurun_out/prof_x
done
