"""Host side of main.py's iteration (select_action -> add -> train, no environment): per-part wall times with the device idle between parts
(each part followed by a synchronize), and a cProfile of the un-synchronised loop.
    python tools/exp/host_loop.py [workload]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
import numpy as np, torch, bench
name = sys.argv[1] if len(sys.argv) > 1 else 'vlsac_halfcheetah_f256_b256'
alg, S, A, B, kw = bench.WORKLOADS[name]
torch.manual_seed(0)
agent = bench.make_agent(alg, S, A, B, kw)
buf, _ = bench.synth_buffer(S, A, 0)
zs = np.zeros(S, np.float32)
for _ in range(100):
    act = agent.select_action(zs, explore=True); buf.add(zs, act, zs, 0.0, 0.0); agent.train(buf, B)
agent.flush(); torch.cuda.synchronize()
n = 1000
t0 = time.perf_counter()
for _ in range(n):
    act = agent.select_action(zs, explore=True); buf.add(zs, act, zs, 0.0, 0.0); agent.train(buf, B)
agent.flush(); torch.cuda.synchronize()
print(f'{name}: {1e6 * (time.perf_counter() - t0) / n:.1f} us per iteration')
parts = dict(select=0.0, add=0.0, train_host=0.0, train_dev=0.0)
for _ in range(n):
    t = time.perf_counter(); act = agent.select_action(zs, explore=True); parts['select'] += time.perf_counter() - t
    t = time.perf_counter(); buf.add(zs, act, zs, 0.0, 0.0); parts['add'] += time.perf_counter() - t
    t = time.perf_counter(); agent.train(buf, B); parts['train_host'] += time.perf_counter() - t
    t = time.perf_counter(); agent.flush(); torch.cuda.synchronize(); parts['train_dev'] += time.perf_counter() - t
print({k: round(1e6 * v / n, 1) for k, v in parts.items()}, 'us per iteration (train_dev = what is left of the device work when train() returns)')
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(n):
    act = agent.select_action(zs, explore=True); buf.add(zs, act, zs, 0.0, 0.0); agent.train(buf, B)
pr.disable(); agent.flush(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(28)
