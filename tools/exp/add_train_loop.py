"""train() rate when every call is preceded by a ReplayBuffer.add() (the shape of main.py's loop without the env step and select_action):
the ring write of call t must wait for the feature chain of call t-1, which samples from the ring on its own stream."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
import numpy as np, torch, bench
alg, S, A, B, kw = bench.WORKLOADS['vlsac_halfcheetah_f256_b256']
agent = bench.make_agent(alg, S, A, B, kw)
buf, _ = bench.synth_buffer(S, A, 0)
s, a = np.zeros(S, np.float32), np.zeros(A, np.float32)
for _ in range(300):
    buf.add(s, a, s, 0.0, 0.0); agent.train(buf, B)
agent.flush(); torch.cuda.synchronize()
for rep in range(3):
    n = 2000
    t0 = time.perf_counter()
    for _ in range(n):
        buf.add(s, a, s, 0.0, 0.0); agent.train(buf, B)
    agent.flush(); torch.cuda.synchronize()
    print(f'add + train: {n / (time.perf_counter() - t0):.0f} train()/s')
