import os, sys, time
os.environ['RLREP_PIPELINE'] = '2'
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
import torch
import bench
alg, S, A, B, kw = bench.WORKLOADS['vlsac_halfcheetah_f256_b256']
agent = bench.make_agent(alg, S, A, B, kw)
buf, _ = bench.synth_buffer(S, A, 0)
for _ in range(30):
    agent.train(buf, B)
agent.flush(); torch.cuda.synchronize()
P = agent._pipe
sa, sb = P['s_ca'], P['s_f']
def t(fn, n=300):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
def feat_only():
    with torch.cuda.stream(sb): P['fs'][0].replay()
def ca_only():
    with torch.cuda.stream(sa): P['ca'][0].replay()
def both_nowait():
    with torch.cuda.stream(sb): P['fs'][0].replay()
    with torch.cuda.stream(sa): P['ca'][0].replay()
def both_wait():
    with torch.cuda.stream(sb):
        P['feat'].replay(); sb.wait_event(P['ev_ca']); P['snap'].replay(); P['ev_snap'].record(sb)
    with torch.cuda.stream(sa):
        sa.wait_event(P['ev_snap']); P['ca'][0].replay(); P['ev_ca'].record(sa)
def both_samestream():
    with torch.cuda.stream(sb): P['fs'][0].replay(); P['ca'][0].replay()
print('feature graph + snapshot alone      %.1f us' % t(feat_only))
print('critic+actor graph alone            %.1f us' % t(ca_only))
print('both, one stream                    %.1f us' % t(both_samestream))
print('both, two streams, no events        %.1f us' % t(both_nowait))
print('both, two streams, with the events  %.1f us' % t(both_wait))
print('agent.train()                       %.1f us' % t(lambda: agent.train(buf, B)))
# priorities: find a high-priority stream that is concurrent with sa, use it for the feature chain (and the reverse)
import itertools
def concurrent(a, b):
    def one():
        with torch.cuda.stream(a): P['ca'][0].replay(); P['ca'][0].replay()
    def two():
        with torch.cuda.stream(a): P['ca'][0].replay()
        with torch.cuda.stream(b): P['ca'][0].replay()
    return t(two, 20) < 0.8 * t(one, 20)
hi = [torch.cuda.Stream(priority=-1) for _ in range(6)]
for name, fa, fb in (('feature chain high priority', sa, None), ('critic/actor chain high priority', None, sb)):
    for h in hi:
        a, b = (fa or h), (fb or h)
        if not concurrent(a, b):
            continue
        def both(a=a, b=b):
            with torch.cuda.stream(b):
                P['feat'].replay(); b.wait_event(P['ev_ca']); P['snap'].replay(); P['ev_snap'].record(b)
            with torch.cuda.stream(a):
                a.wait_event(P['ev_snap']); P['ca'][0].replay(); P['ev_ca'].record(a)
        print('%-34s %.1f us' % (name, t(both)))
        break
