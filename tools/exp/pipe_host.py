"""Host issue time vs total time per train() of the single-GPU pipelined form (is the loop host-bound?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
import torch
import bench
alg, S, A, B, kw = bench.WORKLOADS['vlsac_halfcheetah_f256_b256']
torch.manual_seed(0)
agent = bench.make_agent(alg, S, A, B, kw)
buf, _ = bench.synth_buffer(S, A, 0)
for _ in range(200): agent.train(buf, B)
agent.flush(); torch.cuda.synchronize()
for n in (50, 200, 1000, 3000):
    t0 = time.perf_counter()
    for _ in range(n): agent.train(buf, B)
    t1 = time.perf_counter()
    agent.flush(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'n={n}: host issue {1e6 * (t1 - t0) / n:.1f} us per train(), total {1e6 * (t2 - t0) / n:.1f} us per train()')
# GPU time of each chain alone (graph replays on its stream, the other chain idle)
P = agent._pipe
if P and P.get('mode') == 2:
    for name, graphs, stream in (('feature chain', P['fs'], P['s_f']), ('critic/actor chain', P['ca'], P['s_ca'])):
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            t0 = time.perf_counter()
            for i in range(500): graphs[i & 1].replay()
            stream.synchronize()
            t1 = time.perf_counter()
        print(f'{name} alone: {1e6 * (t1 - t0) / 500:.1f} us per replay')

# What stretches the feature chain when something else runs beside it?  (a) 25 tiny launches, (b) 4 chip-filling ~20 us launches,
# (c) both, each as a graph replayed on the critic/actor stream beside the real feature-chain graph.
if P and P.get('mode') == 2:
    c = agent.core
    small = torch.empty(256, device='cuda'); big = torch.empty(24 << 20, device='cuda')
    def cap(fn):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=P['s_ca']):
            fn()
        return g
    def tiny(n=25):
        for _ in range(n): c.fill_normal(small, 1.0, 1, 2)
    def heavy(n=4):
        for _ in range(n): c.fill_normal(big, 1.0, 1, 2)
    cases = {'25 tiny launches': cap(tiny), '4 chip-filling launches': cap(heavy), 'both': cap(lambda: (tiny(), heavy())),
             '50 tiny launches': cap(lambda: tiny(50)), '8 chip-filling launches': cap(lambda: heavy(8))}
    for name, g in cases.items():
        torch.cuda.synchronize()
        with torch.cuda.stream(P['s_ca']):
            t0 = time.perf_counter()
            for i in range(300): g.replay()
            P['s_ca'].synchronize()
            alone = 1e6 * (time.perf_counter() - t0) / 300
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(300):
            with torch.cuda.stream(P['s_f']): P['fs'][i & 1].replay()
            with torch.cuda.stream(P['s_ca']): g.replay()
        torch.cuda.synchronize()
        both = 1e6 * (time.perf_counter() - t0) / 300
        print(f'{name}: alone {alone:.1f} us; beside the feature chain (304.8 alone): {both:.1f} us per iteration')

if P and P.get('mode') == 2:
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(500):
        with torch.cuda.stream(P['s_f']): P['fs'][i & 1].replay()
        with torch.cuda.stream(P['s_ca']): P['ca'][i & 1].replay()
    torch.cuda.synchronize()
    print(f'both chains on the chosen pair, no events: {1e6 * (time.perf_counter() - t0) / 500:.1f} us per iteration')
    ev = [torch.cuda.Event() for _ in range(4)]
    t0 = time.perf_counter()
    for i in range(500):
        k = i & 1
        with torch.cuda.stream(P['s_f']):
            if i >= 2: P['s_f'].wait_event(ev[2 + k])
            P['fs'][k].replay(); ev[k].record(P['s_f'])
        with torch.cuda.stream(P['s_ca']):
            P['s_ca'].wait_event(ev[k]); P['ca'][k].replay(); ev[2 + k].record(P['s_ca'])
    torch.cuda.synchronize()
    print(f'both chains on the chosen pair, with the snapshot / reuse events: {1e6 * (time.perf_counter() - t0) / 500:.1f} us per iteration')


# Which heavy launch of the critic / actor chain costs the feature chain how much?  Each one alone, 4x per graph, beside the feature chain.
if P and P.get('mode') == 2:
    c = agent.core
    heavy = []
    for prog in (2, 4):
        for i, n in enumerate(c.stages(prog)):
            if n.startswith('noise critic'):
                heavy.append((prog, i, n))
    t0 = time.perf_counter()
    for i in range(300): 
        with torch.cuda.stream(P['s_f']): P['fs'][i & 1].replay()
    torch.cuda.synchronize()
    f_alone = 1e6 * (time.perf_counter() - t0) / 300
    for prog, i, n in heavy:
        for _ in range(3): c.run_stage(prog, i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=P['s_ca']):
            for _ in range(4): c.run_stage(prog, i)
        torch.cuda.synchronize()
        with torch.cuda.stream(P['s_ca']):
            t0 = time.perf_counter()
            for _ in range(300): g.replay()
            P['s_ca'].synchronize()
            alone = 1e6 * (time.perf_counter() - t0) / 300
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(300):
            with torch.cuda.stream(P['s_f']): P['fs'][k & 1].replay()
            with torch.cuda.stream(P['s_ca']): g.replay()
        torch.cuda.synchronize()
        both = 1e6 * (time.perf_counter() - t0) / 300
        print(f'4 x [{n}]: {alone:.1f} us alone; feature chain {f_alone:.1f} -> {both:.1f} us beside it = +{(both - f_alone) / alone:.2f} us per us')

if P and P.get('mode') == 2:
    # the rest of the critic / actor programs (everything but the four noise-critic launches), as one graph beside the feature chain
    c = agent.core
    rest = [(prog, i) for prog in (2, 3, 4, 5) for i, n in enumerate(c.stages(prog)) if not n.startswith('noise critic')]
    for prog, i in rest: c.run_stage(prog, i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=P['s_ca']):
        for prog, i in rest: c.run_stage(prog, i)
    torch.cuda.synchronize()
    with torch.cuda.stream(P['s_ca']):
        t0 = time.perf_counter()
        for _ in range(300): g.replay()
        P['s_ca'].synchronize()
        alone = 1e6 * (time.perf_counter() - t0) / 300
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(300):
        with torch.cuda.stream(P['s_f']): P['fs'][k & 1].replay()
        with torch.cuda.stream(P['s_ca']): g.replay()
    torch.cuda.synchronize()
    both = 1e6 * (time.perf_counter() - t0) / 300
    print(f'{len(rest)} small launches of the critic / actor programs: {alone:.1f} us alone; feature chain {f_alone:.1f} -> {both:.1f} us beside them')
