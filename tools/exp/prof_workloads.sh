# kernel-time breakdown of the other BASELINE workloads (rocprofv3 --kernel-trace --stats), top 20 kernels each -> gpurun_out/profiles_r03/
export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/profiles_r03
for w in spedersac_ant_f512_b1024 ctrlsac_halfcheetah_f2048_b256 ctrlsac_halfcheetah_f256_b256 sac_halfcheetah_b256 diffsrsac_humanoid_b2048; do
  st=200; [ $w = diffsrsac_humanoid_b2048 ] && st=10
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_x -- python3 $R/bench.py --workload $w --steps $st --warmup 5 --no-cpu --no-profile > $R/gpurun_out/prof_x.log 2>&1
  f=$(ls $R/gpurun_out/prof_x/*/*kernel_stats.csv | head -1)
  python3 - "$f" "$R/gpurun_out/profiles_r03/r03_${w}_kernel_stats.csv" <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'at::native' not in r['Name'] and 'rocclr' not in r['Name']]
with open(sys.argv[2],'w') as f:
    w=csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows[:20])
print(sys.argv[2], len(rows))
PY
  rm -rf $R/gpurun_out/prof_x
done
