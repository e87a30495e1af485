export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 400 python3 -m pytest tests/test_comm.py -x -q -m gpu > gpurun_out/pytest_comm.log 2>&1; rc=$?; tail -n 25 gpurun_out/pytest_comm.log
exit $rc
