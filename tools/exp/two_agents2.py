import os, sys, time
os.environ['RLREP_PIPELINE'] = '0'
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
import torch
import bench
from rlrep_amd.agent.sac import sac_agent
alg, S, A, B, kw = bench.WORKLOADS['vlsac_halfcheetah_f256_b256']
agents = [bench.make_agent(alg, S, A, B, kw) for _ in range(2)]
bufs = [bench.synth_buffer(S, A, i)[0] for i in range(2)]
for a, b_ in zip(agents, bufs):
    for _ in range(20):
        a.train(b_, B)
torch.cuda.synchronize()
def run(streams, n_agents, n=300):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        for k in range(n_agents):
            with torch.cuda.stream(streams[k]):
                agents[k]._graph.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
cands = [torch.cuda.Stream() for _ in range(8)]
one = run([cands[0]], 1)
print('one agent', one)
for j in range(1, 8):
    two = run([cands[0], cands[j]], 2)
    print(f'streams 0,{j}: two agents {two:.1f} us  (x{2 * one / two:.2f})')
pair = sac_agent._concurrent_stream_pair(agents[0].core)
two = run(list(pair), 2)
print(f'picked pair: two agents {two:.1f} us (x{2 * one / two:.2f})')
