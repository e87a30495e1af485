"""Do dependent launches of DIFFERENT kernels cost more than repeats of one?  (instruction-cache / cold-operand effects)
Stage i repeated 60x in a graph vs stages cycled (i, j, k, ...) 60 launches in a graph; feature program of the headline config."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
import torch
import bench
alg, S, A, B, kw = bench.WORKLOADS['vlsac_halfcheetah_f256_b256']
agent = bench.make_agent(alg, S, A, B, kw)
buf, _ = bench.synth_buffer(S, A, 0)
for _ in range(20): agent.train(buf, B)
agent.flush(); torch.cuda.synchronize()
core = agent.core
names = core.stages(0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timed(seq, reps=60):
    n = 0
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    for i in seq: core.run_stage(0, i)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        while n < reps:
            for i in seq:
                core.run_stage(0, i); n += 1
    g.replay(); torch.cuda.synchronize()
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * n)
single = {i: timed([i]) for i in range(len(names))}
for i, n in enumerate(names): print(f'{i:2d} {single[i]:6.2f} us  {n}')
allseq = list(range(len(names)))
t = timed(allseq)
print(f'all {len(names)} stages cycled in program order: {t:.2f} us per launch; mean of the single-stage loops {sum(single.values()) / len(single):.2f}')
for seq in ([0, 1], [0, 9], [1, 6], [4, 8], [0, 1, 2], [6, 7, 8, 9]):
    t = timed(seq)
    print(f'cycle {seq}: {t:.2f} us per launch; singles mean {sum(single[i] for i in seq) / len(seq):.2f}')
