export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_pipe -- python3 $R/bench.py --steps 400 --warmup 20 --no-cpu --no-profile --quick > $R/gpurun_out/trace_pipe.log 2>&1 || { tail -5 $R/gpurun_out/trace_pipe.log; exit 1; }
python3 $R/tools/exp/trace_overlap.py $R/gpurun_out/trace_pipe > $R/gpurun_out/trace_overlap.txt 2>&1
cat $R/gpurun_out/trace_overlap.txt
python3 $R/tools/exp/trace_gaps.py $R/gpurun_out/trace_pipe > $R/gpurun_out/trace_gaps.txt 2>&1
rm -rf $R/gpurun_out/trace_pipe
