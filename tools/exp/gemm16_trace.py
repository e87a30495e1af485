"""Which front end every gemm16 launch of one train() gets (RLREP_ENABLE=gemm16_trace makes the launcher print one line per launch):
python tools/exp/gemm16_trace.py [workload]   -- eager train() (stage by stage), the third call traced."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ['RLREP_ENABLE'] = 'gemm16_trace'
import torch
import bench

w = sys.argv[1] if len(sys.argv) > 1 else 'vlsac_halfcheetah_f256_b256'
alg, S, A, B, kw = bench.WORKLOADS[w]
agent = bench.make_agent(alg, S, A, B, dict(kw, graph=False, pipeline=False))
buf, _ = bench.synth_buffer(S, A, 1)
for t in range(3):
    if t == 2:
        sys.stderr.write('==== traced train() ====\n')
    agent.train(buf, B)
    torch.cuda.synchronize()
    if t < 2:
        sys.stderr.write('---- (untraced call above) ----\n')
