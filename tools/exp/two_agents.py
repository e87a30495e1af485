"""Do two INDEPENDENT train() launch chains overlap on one GPU?  Two vlsac agents (sequential graph mode) on two streams against one
agent alone: separates 'our kernels do not overlap' from 'the deferred pipeline serialises somewhere'."""
import os, sys, time
os.environ['RLREP_PIPELINE'] = '0'
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
import torch
import bench
alg, S, A, B, kw = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else 'vlsac_halfcheetah_f256_b256']
agents = [bench.make_agent(alg, S, A, B, kw) for _ in range(2)]
if os.environ.get('NO_PREFETCH_BATCH'):
    for a in agents:
        a.core.prefetch_batch = lambda *x: False
if os.environ.get('NO_POLICY_PREFETCH'):
    for a in agents:
        a.core.prefetch_policy_early = lambda *x: False
        a.core.prefetch_policy = lambda *x: False
bufs = [bench.synth_buffer(S, A, i)[0] for i in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for a, b_ in zip(agents, bufs):
    for _ in range(20):
        a.train(b_, B)
torch.cuda.synchronize()
def run(n_agents, n=500):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        for k in range(n_agents):
            with torch.cuda.stream(streams[k]):
                agents[k].train(bufs[k], B)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
for k in (1, 2, 1, 2):
    us = run(k)
    print(f'{k} agent(s): {us:.1f} us per round = {k * 1e6 / us:.0f} train()/s aggregate')
# the same with the bare graph replays (no per-call metric snapshot / Python in between)
def run_bare(n_agents, n=500, clone=False):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        for k in range(n_agents):
            with torch.cuda.stream(streams[k]):
                agents[k]._graph.replay()
                if clone:
                    agents[k].core.metrics_tensor().clone()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
for clone in (False, True):
    for k in (1, 2):
        us = run_bare(k, clone=clone)
        print(f'bare replay, clone={clone}, {k} agent(s): {us:.1f} us per round = {k * 1e6 / us:.0f} train()/s aggregate')
# my own capture of the agent's _body (same content as agent._graph, captured here)
cap = torch.cuda.Stream()
mine = []
for k in range(2):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cap):
        agents[k]._body(bufs[k], B, True)
    mine.append(g)
def run_mine(n_agents, n=500, gs=None):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        for k in range(n_agents):
            with torch.cuda.stream(streams[k]):
                gs[k].replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
for k in (1, 2):
    print(f'own capture of _body, {k} agent(s): {run_mine(k, gs=mine):.1f} us per round')
# two captures of the SAME agent's body on two streams would race on its buffers; instead: agent graphs launched from ONE stream vs two
s0 = torch.cuda.Stream()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(500):
    with torch.cuda.stream(s0):
        agents[0]._graph.replay(); agents[1]._graph.replay()
torch.cuda.synchronize()
print(f'both agent graphs on ONE stream: {(time.perf_counter() - t0) / 500 * 1e6:.1f} us per round')
# bisect: custom bodies from the ABI entry points
F = kw['feature_dim']
eps_f = [torch.randn(B, F, device='cuda') for _ in range(2)]
eps_a = [torch.randn(B, A, device='cuda') for _ in range(2)]
def body(k, level):
    c = agents[k].core
    if level >= 4: c.begin_train()
    if level >= 5: c.sample(0, bufs[k].ring, torch.zeros(B, dtype=torch.int32, device='cuda') if False else idxs[k], B)
    for _ in range(4):
        c.feature_step(eps_f[k])
    if level >= 2:
        c.critic_step(eps_a[k]); c.actor_step(eps_a[k])
    if level >= 3: c.update_target()
idxs = [torch.randint(0, 65536, (B,), dtype=torch.int32, device='cuda') for _ in range(2)]
for level in (1, 2, 3, 4, 5):
    gs = []
    for k in range(2):
        body(k, level); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=cap):
            body(k, level)
        gs.append(g)
    one, two = run_mine(1, 200, gs), run_mine(2, 200, gs)
    print(f'body level {level}: one {one:.1f} us, two {two:.1f} us  (x{2 * one / two:.2f} overlap)')
# same agents, same harness: run_stage-built body vs ABI-built body
def body_rs(k, progs):
    c = agents[k].core
    for p in progs:
        for i in range(len(c.stages(p))):
            c.run_stage(p, i)
def body_abi(k, what):
    c = agents[k].core
    if what == 'fb':
        c.feature_backward(eps_f[k])
    elif what == 'fa':
        c.feature_apply()
    elif what == 'fs':
        c.feature_step(eps_f[k])
for name, fn in (('run_stage prog 0', lambda k: body_rs(k, (0,))), ('ABI feature_backward', lambda k: body_abi(k, 'fb')),
                 ('run_stage prog 1', lambda k: body_rs(k, (1,))), ('ABI feature_apply', lambda k: body_abi(k, 'fa')),
                 ('run_stage prog 0+1', lambda k: body_rs(k, (0, 1))), ('ABI feature_step', lambda k: body_abi(k, 'fs'))):
    gs = []
    for k in range(2):
        fn(k); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=cap):
            for _ in range(4):
                fn(k)
        gs.append(g)
    one, two = run_mine(1, 200, gs), run_mine(2, 200, gs)
    print(f'{name:24s}: one {one:.1f} us, two {two:.1f} us  (x{2 * one / two:.2f} overlap)')
