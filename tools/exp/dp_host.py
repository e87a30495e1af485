"""Host issue time vs device time of the data-parallel train() forms over a one-rank RCCL group (RLREP_FORCE_DP=1)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
os.environ['RLREP_FORCE_DP'] = '1'
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
import torch, torch.distributed as dist
import bench
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
alg, S, A, B, kw = bench.WORKLOADS['vlsac_halfcheetah_f256_b256']
torch.manual_seed(0)
agent = bench.make_agent(alg, S, A, B, kw)
buf, _ = bench.synth_buffer(S, A, 0)
for _ in range(100): agent.train(buf, B)
agent.flush(); torch.cuda.synchronize()
n = 1000
t0 = time.perf_counter()
for _ in range(n): agent.train(buf, B)
t1 = time.perf_counter()
agent.flush(); torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'pipeline_dp={os.environ.get("RLREP_PIPELINE_DP", "1")}: host issue {1e6 * (t1 - t0) / n:.1f} us per train(), total {1e6 * (t2 - t0) / n:.1f} us per train()')
# the bare collective: host cost and device time of one all_reduce of the feature gradient slice
lay = agent.core.layout
view = agent.core.grads[lay.group_offset[0]:lay.group_offset[0] + lay.group_floats[0]]
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(1000): dist.all_reduce(view)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f'all_reduce({view.numel() * 4 / 1e6:.2f} MB, one rank): host {1e6 * (t1 - t0) / 1000:.1f} us, total {1e6 * (t2 - t0) / 1000:.1f} us per call')
dist.destroy_process_group()
