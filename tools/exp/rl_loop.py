"""main.py's loop without the environment: select_action (needs the finished actor: a flush + a device sync), add, train -- per iteration.
    python tools/exp/rl_loop.py            RLREP_PIPELINE=0 python tools/exp/rl_loop.py"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
import numpy as np, torch, bench
wl = sys.argv[1] if len(sys.argv) > 1 else 'vlsac_halfcheetah_f256_b256'
alg, S, A, B, kw = bench.WORKLOADS[wl]
agent = bench.make_agent(alg, S, A, B, kw)
buf, _ = bench.synth_buffer(S, A, 0)
s = np.zeros(S, np.float32)
for _ in range(300):
    a = agent.select_action(s, explore=True); buf.add(s, a, s, 0.0, 0.0); agent.train(buf, B)
agent.flush(); torch.cuda.synchronize()
for rep in range(3):
    n = 1500
    t0 = time.perf_counter()
    for _ in range(n):
        a = agent.select_action(s, explore=True); buf.add(s, a, s, 0.0, 0.0); agent.train(buf, B)
    agent.flush(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'{wl}: select_action + add + train: {n / dt:.0f} iterations/s ({1e6 * dt / n:.0f} us each)')
