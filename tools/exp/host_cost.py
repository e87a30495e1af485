"""Host cost per iteration of the two-stream pipeline choreography (3 graph replays + 2 event records + 2 waits) with trivial graphs."""
import time, torch
x = torch.zeros(64, device='cuda')
gs = []
s = torch.cuda.Stream()
for _ in range(3):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        x.add_(1.0)
    gs.append(g)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
ea, eb = torch.cuda.Event(), torch.cuda.Event()
torch.cuda.synchronize()
for mode in ('two-stream', 'one-graph'):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 3000
        for i in range(n):
            if mode == 'two-stream':
                with torch.cuda.stream(sb):
                    gs[0].replay()
                    sb.wait_event(ea)
                    gs[1].replay()
                    eb.record(sb)
                with torch.cuda.stream(sa):
                    sa.wait_event(eb)
                    gs[2].replay()
                    ea.record(sa)
            else:
                gs[0].replay()
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f'{mode}: host {1e6 * (t1 - t0) / n:.1f} us per iteration (+ {1e3 * (t2 - t1):.2f} ms to drain)')
