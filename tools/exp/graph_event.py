"""Event record after a graph launch (stream op) vs as the graph's last node (external event): cost on a chain of short kernels, and does
a waiter on another stream see it?"""
import time, torch
x = torch.zeros(1 << 16, device='cuda'); y = torch.zeros(1 << 16, device='cuda')
s = torch.cuda.Stream(); s2 = torch.cuda.Stream()
def chain(n=20):
    for _ in range(n): x.add_(1.0)
res = {}
for mode in ('stream-record', 'graph-record', 'no-record'):
    try:
        ev = torch.cuda.Event(external=True) if mode == 'graph-record' else torch.cuda.Event()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            chain()
            if mode == 'graph-record': ev.record()
        torch.cuda.synchronize()
        def it():
            with torch.cuda.stream(s):
                g.replay()
                if mode == 'stream-record': ev.record(s)
            if mode != 'no-record':
                with torch.cuda.stream(s2):
                    s2.wait_event(ev); y.add_(1.0)
        for _ in range(20): it()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(500): it()
        torch.cuda.synchronize()
        res[mode] = 1e6 * (time.perf_counter() - t0) / 500
        print(f'{mode}: {res[mode]:.1f} us per iteration (20-kernel graph + waiter)')
    except Exception as e:
        print(mode, 'failed:', type(e).__name__, str(e)[:200])
