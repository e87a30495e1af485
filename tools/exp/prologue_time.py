"""What the train() prologue launch costs, alone (graph of 200 launches, HIP events), as a function of the noise it generates."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
import torch, bench
alg, S, A, B, kw = bench.WORKLOADS['vlsac_halfcheetah_f256_b256']
agent = bench.make_agent(alg, S, A, B, kw)
buf, _ = bench.synth_buffer(S, A, 0)
for _ in range(5): agent.train(buf, B)
agent.flush(); torch.cuda.synchronize()
c = agent.core
idx = torch.zeros(5 * B, dtype=torch.int32, device='cuda')
full = 4 * B * 256 + 2 * B * A
for n_eps in (full, B * 256, 1024):
    eps = torch.zeros(full, device='cuda')
    s = torch.cuda.Stream(); g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        c.train_prologue(buf.ring, buf.size_dev(), idx, eps[:n_eps], 1, 0, 1 << 40, B)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for _ in range(200):
            c.train_prologue(buf.ring, buf.size_dev(), idx, eps[:n_eps], 1, 0, 1 << 40, B)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(f'n_eps = {n_eps}: {e0.elapsed_time(e1) * 1000 / 200:.2f} us per prologue launch (back to back)')
