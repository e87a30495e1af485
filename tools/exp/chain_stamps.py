"""Device-side dates of the two launch chains of the pipelined train() (no profiler): a one-thread stamp launch (rlrep_debug_stamp,
100 MHz wall clock) is captured at the head and tail of the feature graph and of the critic/actor graph (RLREP_ENABLE=stamp, read by
SACAgent).  Prints, over the last calls: chain lengths, the idle time of the feature queue between two calls, the period.
    RLREP_ENABLE=stamp python tools/exp/chain_stamps.py [workload]"""
import os, sys
os.environ.setdefault('RLREP_ENABLE', 'stamp')
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
import numpy as np, torch
import bench
alg, S, A, B, kw = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else 'vlsac_halfcheetah_f256_b256']
torch.manual_seed(0)
agent = bench.make_agent(alg, S, A, B, kw)
buf, _ = bench.synth_buffer(S, A, 0)
agent._stamp_ring = torch.zeros(1 + 8192, dtype=torch.int64, device='cuda')     # before any capture
N = 600
for _ in range(N): agent.train(buf, B)
agent.flush(); torch.cuda.synchronize()
ring = agent._stamp_ring.cpu().numpy()
n, cap = int(ring[0]), len(ring) - 1
vals = ring[1:1 + min(n, cap)]
t, tag = (vals >> 8).astype(np.int64), (vals & 255).astype(np.int64)
o = np.argsort(t, kind='stable'); t, tag = t[o], tag[o]
ev = {k: t[tag == k] / 100.0 for k in (1, 2, 3, 4)}     # us
m = min(len(v) for v in ev.values())
f0, f1, c0, c1 = (ev[k][-m:][-200:] for k in (1, 2, 3, 4))
print(f'{len(f0)} calls: period {np.median(np.diff(f0)):.1f} us')
print(f'feature chain (head stamp -> tail stamp) median {np.median(f1 - f0):.1f} us   p90 {np.percentile(f1 - f0, 90):.1f}')
print(f'feature queue idle: tail(t) -> head(t+1) median {np.median(f0[1:] - f1[:-1]):.1f} us   p90 {np.percentile(f0[1:] - f1[:-1], 90):.1f}')
print(f'critic/actor chain median {np.median(c1 - c0):.1f} us   start after feature tail by {np.median(c0 - f1):.1f} us')
print(f'critic/actor queue idle: tail(t) -> head(t+1) median {np.median(c0[1:] - c1[:-1]):.1f} us')
