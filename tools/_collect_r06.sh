# Collect everything profiles/r06_* is built from (run on the GPU box: gpurun -- 'bash tools/_collect_r06.sh [part]').  Parts keep a call inside
# gpurun's time limit: pmc | stats | bench | misc.  Summaries are written ON the box (raw rocprofv3 trees exceed what gpurun_out/ carries back).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profiles_r06; mkdir -p $OUT
PART=${1:-all}
cd /tmp
pmc_one() {   # workload, train() calls [, "graph": the passes run the TIMED form (hipGraph replay, two chains) instead of the eager one]
  w=$1; n=$2; NOGRAPH=--no-graph; [ "$3" = graph ] && NOGRAPH=""
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq_$w -- python3 $R/bench.py --workload $w --steps $n --warmup 5 --no-cpu $NOGRAPH --no-profile --quick > $R/gpurun_out/pmc_sq_$w.log 2>&1 || return 1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch_$w -- python3 $R/bench.py --workload $w --steps $n --warmup 5 --no-cpu $NOGRAPH --no-profile --quick > $R/gpurun_out/pmc_fetch_$w.log 2>&1 || return 1
  rocprofv3 --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write_$w -- python3 $R/bench.py --workload $w --steps $n --warmup 5 --no-cpu $NOGRAPH --no-profile --quick > $R/gpurun_out/pmc_write_$w.log 2>&1 || return 1
  calls=$(grep -o '"total_train_calls": [0-9]*' $R/gpurun_out/pmc_fetch_$w.log | tail -n 1 | grep -o '[0-9]*$')
  RLREP_PMC_FORM=${3:-eager} RLREP_PROFILES_OUT=$OUT python3 $R/tools/summarize_pmc.py r06 $w ${calls:-0} || return 1
  cp $OUT/r06_pmc_*.json $R/profiles/                       # (bench.py on this box reads roofline.traffic from profiles/)
  rm -rf $R/gpurun_out/pmc_sq_$w $R/gpurun_out/pmc_fetch_$w $R/gpurun_out/pmc_write_$w
  echo "[collect] pmc $w done ($calls train() calls)"
}
stats_one() {  # workload, steps
  w=$1; st=$2
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_x -- python3 $R/bench.py --workload $w --steps $st --warmup 5 --no-cpu --no-profile --quick > $R/gpurun_out/prof_$w.log 2>&1 || return 1
  f=$(ls $R/gpurun_out/prof_x/*/*kernel_stats.csv | head -1)
  calls=$(grep -o '"total_train_calls": [0-9]*' $R/gpurun_out/prof_$w.log | tail -n 1 | grep -o '[0-9]*$')
  python3 - "$f" "$OUT/r06_${w}_kernel_stats.csv" "${calls:-0}" <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'at::native' not in r['Name'] and 'rocclr' not in r['Name']]
with open(sys.argv[2],'w') as f:
    f.write(f'# train() calls in this run: {sys.argv[3]}  (Calls / that = launches per train())\n')
    w=csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows[:28])
print(sys.argv[2], len(rows))
PY
  rm -rf $R/gpurun_out/prof_x
  echo "[collect] stats $w done"
}
if [ $PART = pmc ] || [ $PART = pmc1 ] || [ $PART = all ]; then
  pmc_one vlsac_halfcheetah_f256_b256 20 graph || exit 1
  pmc_one ctrlsac_halfcheetah_f2048_b256 10 || exit 1
  pmc_one spedersac_ant_f512_b1024 10 || exit 1
  pmc_one ctrlsac_halfcheetah_f256_b256 20 || exit 1
fi
if [ $PART = pmc ] || [ $PART = pmc2 ] || [ $PART = all ]; then
  pmc_one sac_halfcheetah_b256 20 || exit 1
  pmc_one sac_pendulum_b64 20 || exit 1
  pmc_one diffsrsac_halfcheetah_b256 10 || exit 1
  pmc_one diffsrsac_humanoid_b2048 3 || exit 1
fi
if [ $PART = pmc_head ]; then
  pmc_one vlsac_halfcheetah_f256_b256 20 graph || exit 1
fi
if [ $PART = stats ] || [ $PART = all ]; then
  stats_one vlsac_halfcheetah_f256_b256 300 || exit 1
  stats_one ctrlsac_halfcheetah_f2048_b256 200 || exit 1
  stats_one ctrlsac_halfcheetah_f256_b256 200 || exit 1
  stats_one spedersac_ant_f512_b1024 200 || exit 1
  stats_one sac_halfcheetah_b256 200 || exit 1
  stats_one diffsrsac_humanoid_b2048 10 || exit 1
fi
if [ $PART = bench ] || [ $PART = all ]; then
  cd $R
  python3 bench.py > gpurun_out/bench_r06.log 2>&1 || exit 1
  tail -n 1 gpurun_out/bench_r06.log > $OUT/r06_bench.json
  python3 bench.py --steps 20 --warmup 5 > gpurun_out/bench_r06_driver.log 2>&1 || exit 1
  tail -n 1 gpurun_out/bench_r06_driver.log > $OUT/r06_bench_driver_form.json
  : > $OUT/r06_bench_all.jsonl
  tail -n 1 gpurun_out/bench_r06_driver.log >> $OUT/r06_bench_all.jsonl
  for w in ctrlsac_halfcheetah_f2048_b256 ctrlsac_halfcheetah_f256_b256 spedersac_ant_f512_b1024 sac_halfcheetah_b256 sac_pendulum_b64 diffsrsac_halfcheetah_b256; do
    python3 bench.py --workload $w --steps 1000 --warmup 100 > gpurun_out/bench_$w.log 2>&1 || exit 1
    tail -n 1 gpurun_out/bench_$w.log >> $OUT/r06_bench_all.jsonl; echo "[collect] bench $w"
  done
  python3 bench.py --workload diffsrsac_humanoid_b2048 --steps 20 --warmup 3 > gpurun_out/bench_diffsrsac_humanoid_b2048.log 2>&1 || exit 1
  tail -n 1 gpurun_out/bench_diffsrsac_humanoid_b2048.log >> $OUT/r06_bench_all.jsonl
fi
if [ $PART = hum ]; then      # diffsrsac Humanoid alone (re-collected after the 256 x 128 bf16x3 tile was routed): PMC first, bench.py reads roofline.traffic from it
  pmc_one diffsrsac_humanoid_b2048 3 || exit 1
  stats_one diffsrsac_humanoid_b2048 10 || exit 1
  cd $R
  python3 bench.py --workload diffsrsac_humanoid_b2048 --steps 20 --warmup 3 > gpurun_out/bench_diffsrsac_humanoid_b2048.log 2>&1 || exit 1
  tail -n 1 gpurun_out/bench_diffsrsac_humanoid_b2048.log > $OUT/r06_bench_humanoid.json
fi
if [ $PART = misc ] || [ $PART = all ]; then
  cd $R
  : > $OUT/r06_dp_rehearsal.jsonl
  RLREP_FORCE_DP=1 python3 bench.py --steps 2000 --warmup 200 --no-cpu --no-profile 2>/dev/null | tail -n 1 >> $OUT/r06_dp_rehearsal.jsonl
  RLREP_FORCE_DP=1 RLREP_DP_CAPTURE=0 python3 bench.py --steps 2000 --warmup 200 --no-cpu --no-profile 2>/dev/null | tail -n 1 >> $OUT/r06_dp_rehearsal.jsonl
  RLREP_FORCE_DP=1 RLREP_PIPELINE_DP=1 python3 bench.py --steps 2000 --warmup 200 --no-cpu --no-profile 2>/dev/null | tail -n 1 >> $OUT/r06_dp_rehearsal.jsonl
  RLREP_ENABLE=stamp python3 tools/exp/chain_stamps.py > $OUT/r06_chain_stamps.txt 2>&1
  # what the in-launch exchange costs with the wire free (one process, one GPU): one attached replica with its peer marked as arrived against a
  # single-GPU agent, and two replicas on the chip attached against independent
  : > $OUT/r06_dp_loopback.txt
  for w in vlsac_halfcheetah_f256_b256 spedersac_ant_f512_b1024 ctrlsac_halfcheetah_f256_b256 sac_halfcheetah_b256; do
    python3 tools/exp/dp_loopback.py --workload $w --arms alone,alone_attached --calls 400 2>/dev/null | grep '^{' >> $OUT/r06_dp_loopback.txt
  done
  for w in vlsac_halfcheetah_f256_b256 spedersac_ant_f512_b1024 ctrlsac_halfcheetah_f256_b256; do
    python3 tools/exp/dp_loopback.py --workload $w --world 2 --calls 300 2>/dev/null | grep '^{' >> $OUT/r06_dp_loopback.txt
  done
  # in-kernel timeline of the 16-row engine's launches (instrumented library: OBJDIR=.obj_tim OUTNAME=librlrep_hip_tim.so EXTRA_FLAGS=-DRL_TIMING bash rlrep_amd/csrc/build.sh)
  [ -f $R/rlrep_amd/lib/librlrep_hip_tim.so ] && RLREP_LIB=$R/rlrep_amd/lib/librlrep_hip_tim.so python3 tools/exp/gemm_timeline.py 2>/dev/null | grep -v amdgpu.ids > $OUT/r06_gemm_timeline.txt
  RLREP_ENABLE=gemm16_trace python3 tools/exp/gemm16_trace.py 2>&1 | awk '/==== traced/{f=1;next} f' | grep gemm16 > $OUT/r06_gemm16_trace.txt
  # two gloo ranks on this one GPU through bench.py's N > 1 path: replicas_identical / allreduce_us_per_train fields (not a scaling number)
  # (a protocol rehearsal, not a scaling number: the two processes time-share the GPU).  Default = gradients summed inside the optimizer launches;
  # RLREP_DP_FUSED=0 = gloo all-reduces between graph segments
  RLREP_DIST_BACKEND=gloo python3 bench.py --gpus 2 --steps 500 --warmup 100 --no-cpu --quick 2>/dev/null | tail -n 1 > $OUT/r06_bench_2ranks_fused_one_gpu.json
  cat $OUT/r06_bench_2ranks_fused_one_gpu.json >> $OUT/r06_dp_rehearsal.jsonl
  RLREP_DP_FUSED=0 RLREP_DIST_BACKEND=gloo python3 bench.py --gpus 2 --steps 30 --warmup 5 --no-cpu --quick 2>/dev/null | tail -n 1 > $OUT/r06_bench_2ranks_gloo_one_gpu.json
  cat $OUT/r06_bench_2ranks_gloo_one_gpu.json >> $OUT/r06_dp_rehearsal.jsonl
fi
echo collected
