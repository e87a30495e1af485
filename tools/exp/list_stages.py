import sys; sys.path[:0]=['/root/repo','/root/repo/tests','/root/repo/tests/golden']
import torch, bench
alg,S,A,B,kw = bench.WORKLOADS['sac_halfcheetah_b256']
agent = bench.make_agent(alg,S,A,B,kw)
buf,_ = bench.synth_buffer(S,A,0)
for _ in range(3): agent.train(buf,B)
for p in (2,7,4,5,3): print(p, agent.core.stages(p))
