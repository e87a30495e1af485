"""Fixed cost of a timed window of K pipelined train() calls (sync; K calls; flush; sync): K = 1, 2, 3, 5, 10, 20 -> slope and intercept."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
import numpy as np, torch, bench
alg, S, A, B, kw = bench.WORKLOADS['vlsac_halfcheetah_f256_b256']
agent = bench.make_agent(alg, S, A, B, kw)
buf, _ = bench.synth_buffer(S, A, 0)
for _ in range(400): agent.train(buf, B)
agent.flush(); torch.cuda.synchronize()
for K in (1, 2, 3, 5, 10, 20, 40):
    ts = []
    for rep in range(15):
        for _ in range(30): agent.train(buf, B)          # busy right up to the window
        agent.flush(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K): agent.train(buf, B)
        t1 = time.perf_counter()
        agent.flush(); torch.cuda.synchronize()
        t2 = time.perf_counter()
        ts.append((1e6 * (t2 - t0), 1e6 * (t1 - t0)))
    a = np.median(np.array(ts), axis=0)
    print(f'K={K:3d}: window {a[0]:8.1f} us (host enqueue {a[1]:7.1f} us)  per call {a[0] / K:7.1f}')
