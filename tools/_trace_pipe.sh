export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT
RLREP_PIPELINE=${MODE:-2} rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_pipe -- python3 $R/bench.py --steps 60 --warmup 20 --no-cpu > $R/gpurun_out/trace_pipe.log 2>&1
echo done
