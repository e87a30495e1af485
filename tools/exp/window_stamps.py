"""Device-side dating (rlrep_debug_stamp) of the FIRST calls of a timed window: sync, K pipelined train() calls, flush.  Shows where a short
window (the driver's --steps 20) loses time against the steady state.   RLREP_ENABLE=stamp python tools/exp/window_stamps.py"""
import os, sys
os.environ.setdefault('RLREP_ENABLE', 'stamp')
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
import numpy as np, torch, time
import bench
alg, S, A, B, kw = bench.WORKLOADS['vlsac_halfcheetah_f256_b256']
agent = bench.make_agent(alg, S, A, B, kw)
buf, _ = bench.synth_buffer(S, A, 0)
agent._stamp_ring = torch.zeros(1 + 8192, dtype=torch.int64, device='cuda')
for _ in range(400): agent.train(buf, B)
agent.flush(); torch.cuda.synchronize()
K = 8
for rep in range(3):
    for _ in range(50): agent.train(buf, B)
    agent.flush(); torch.cuda.synchronize()
    agent._stamp_ring.zero_(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K): agent.train(buf, B)
    m = os.environ.get('WS_MODE', '')
    if m == 'flush_sleep': agent.flush(); time.sleep(0.004)
    if m == 'sync_only': torch.cuda.synchronize()
    if m == 'stream_sync': agent._pipe['s_ca'].synchronize()
    if os.environ.get('WS_SLEEP'): time.sleep(float(os.environ['WS_SLEEP']))          # host idles (no flush, no synchronize) while the tail runs
    if os.environ.get('WS_EVENT'):                                                     # wait for the last chain through its event only
        agent._pipe['ev_ca'][(agent._pipe['t'] - 1) % agent._pipe['nset']].synchronize()
    agent.flush(); torch.cuda.synchronize()
    wall = 1e6 * (time.perf_counter() - t0)
    ring = agent._stamp_ring.cpu().numpy(); n = int(ring[0]); v = ring[1:1 + n]
    t, tag = (v >> 8).astype(np.int64) / 100.0, (v & 255).astype(np.int64)
    o = np.argsort(t, kind='stable'); t, tag = t[o], tag[o]
    ev = {k: t[tag == k] for k in (1, 2, 3, 4)}
    z = ev[1][0]
    print(f'rep {rep}: host window {wall:.0f} us; device first stamp -> last stamp {t[-1] - z:.0f} us')
    for i in range(K):
        print(f'   call {i}: feature {ev[1][i] - z:7.1f} .. {ev[2][i] - z:7.1f} ({ev[2][i] - ev[1][i]:6.1f})   critic/actor {ev[3][i] - z:7.1f} .. {ev[4][i] - z:7.1f} ({ev[4][i] - ev[3][i]:6.1f})'
              + (f'   feature idle before: {ev[1][i] - ev[2][i - 1]:5.1f}' if i else ''))
