"""Can an RCCL all-reduce be captured into a hipGraph through torch.distributed on this stack?  (one-rank group: the only one a one-GPU box allows)"""
import os, time, torch, torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29541')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
x = torch.ones(1 << 19, device='cuda'); y = torch.zeros_like(x)
dist.all_reduce(x); torch.cuda.synchronize()
s = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=s, capture_error_mode='thread_local'):
        y.add_(x)
        dist.all_reduce(y)
        y.mul_(0.5)
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    print('captured and replayed; y[0] =', float(y[0]))
    t0 = time.perf_counter()
    for _ in range(500): g.replay()
    torch.cuda.synchronize()
    print(f'{1e6 * (time.perf_counter() - t0) / 500:.1f} us per replay of (add, all_reduce, mul)')
except Exception as e:
    print('capture failed:', type(e).__name__, str(e)[:300])
dist.destroy_process_group()
