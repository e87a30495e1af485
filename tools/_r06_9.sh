# round 6: soak of the attached forms (two replicas of one process, default two-chain graphs): replicas must stay bit-identical, no wait may run out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out
: > gpurun_out/r06_dp_soak.txt
timeout -k 10 300 python3 tools/exp/dp_loopback.py --workload vlsac_halfcheetah_f256_b256 --arms attached --calls 20000 --warm 200 2>/dev/null | grep '^{' | tee -a gpurun_out/r06_dp_soak.txt
timeout -k 10 300 python3 tools/exp/dp_loopback.py --workload spedersac_ant_f512_b1024 --arms attached --calls 5000 --warm 100 2>/dev/null | grep '^{' | tee -a gpurun_out/r06_dp_soak.txt
timeout -k 10 300 python3 tools/exp/dp_loopback.py --workload ctrlsac_halfcheetah_f256_b256 --arms attached --calls 10000 --warm 100 2>/dev/null | grep '^{' | tee -a gpurun_out/r06_dp_soak.txt
timeout -k 10 300 python3 tools/exp/dp_loopback.py --workload sac_halfcheetah_b256 --arms attached --calls 20000 --warm 100 2>/dev/null | grep '^{' | tee -a gpurun_out/r06_dp_soak.txt
timeout -k 10 300 python3 tools/exp/dp_loopback.py --workload diffsrsac_halfcheetah_b256 --arms attached --calls 3000 --warm 100 2>/dev/null | grep '^{' | tee -a gpurun_out/r06_dp_soak.txt
