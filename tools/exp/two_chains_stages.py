"""Which of OUR kernels' launch chains overlap across two streams?  For several stages: a graph of 50 launches from agent 1 on stream A and
the same from agent 2 on stream B, against one alone."""
import os, sys, time
os.environ['RLREP_PIPELINE'] = '0'
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
import torch
import bench
alg, S, A, B, kw = bench.WORKLOADS['vlsac_halfcheetah_f256_b256']
agents = [bench.make_agent(alg, S, A, B, kw) for _ in range(2)]
bufs = [bench.synth_buffer(S, A, i)[0] for i in range(2)]
for a, b_ in zip(agents, bufs):
    for _ in range(5):
        a.train(b_, B)
torch.cuda.synchronize()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
cap = torch.cuda.Stream()
L = 50
def graphs(fn):
    gs = []
    for k in range(2):
        fn(k); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=cap):
            for _ in range(L):
                fn(k)
        gs.append(g)
    return gs
def timeit(gs, n_agents, n=int(os.environ.get('NREPLAY', 40))):
    for k in range(n_agents):
        with torch.cuda.stream(streams[k]): gs[k].replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        for k in range(n_agents):
            with torch.cuda.stream(streams[k]): gs[k].replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n / L * 1e6
tens = [torch.empty(4096, device='cuda') for _ in range(2)]
names = agents[0].core.stages(0)
cases = [('philox fill (trivial kernel)', lambda k: agents[k].core.fill_normal(tens[k], 1.0, 1, 2))]
idxs = [torch.randint(0, 65536, (B,), dtype=torch.int32, device='cuda') for _ in range(2)]
itens = [torch.empty(1024, dtype=torch.int32, device='cuda') for _ in range(2)]
cases.append(('replay_sample (fill_slot kernel)', lambda k: agents[k].core.sample(0, bufs[k].ring, idxs[k], B)))
cases.append(('begin_train (counter_inc <<<1,64>>>)', lambda k: agents[k].core.begin_train()))
cases.append(('fill_indices_dev (philox, reads step counter)', lambda k: agents[k].core.fill_indices_dev(itens[k], bufs[k].size_dev(), 1, 1 << 40)))
cases.append(('fill_normal_dev', lambda k: agents[k].core.fill_normal_dev(tens[k], 1.0, 1, 2 << 40)))
cases.append(('update_target (folded bracket off)', lambda k: agents[k].core.update_target()))
for prog, idx in ((0, 0),):
    nm = agents[0].core.stages(prog)[idx]
    cases.append((f'stage {prog}/{idx} {nm}', (lambda p, i: (lambda k: agents[k].core.run_stage(p, i)))(prog, idx)))
def prog_fn(p):
    n = len(agents[0].core.stages(p))
    return lambda k: [agents[k].core.run_stage(p, i) for i in range(n)]
if os.environ.get('PROGRAMS'):
    L = int(os.environ.get('REPS', 8))
    cases = [(f'program {p} ({len(agents[0].core.stages(p))} launches)', prog_fn(p)) for p in range(7)]
    def whole(k):
        for p in (0, 1, 0, 1, 0, 1, 0, 1, 2, 3, 4, 5):
            prog_fn(p)(k)
    cases.append(('feature x4 + critic + actor via run_stage', whole))
    cases.append(('agent.train() body (graph off)', None))
for name, fn in cases:
    if fn is None:
        continue
    gs = graphs(fn)
    one, two = timeit(gs, 1), timeit(gs, 2)
    print(f'{name:55s} one chain {one:6.2f} us/launch | two chains {two:6.2f} us per launch pair  (x{2 * one / two:.2f} overlap)')
