"""In-kernel timeline of every gemm16 launch of one graph-replayed train() (needs the -DRL_TIMING build:
   OBJDIR=.obj_tim OUTNAME=librlrep_hip_tim.so EXTRA_FLAGS=-DRL_TIMING bash rlrep_amd/csrc/build.sh; run with RLREP_LIB=rlrep_amd/lib/librlrep_hip_tim.so).
   Times are the 100 MHz wall clock (10 ns ticks) read by thread 0 of each workgroup."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
import numpy as np, torch
import bench
from rlrep_amd import _lib

lib = _lib.lib
raw = C.CDLL(_lib.LIB_PATH)
raw.rl_timing_buffer.argtypes = [C.c_void_p, C.c_uint]; raw.rl_timing_buffer.restype = C.c_int
raw.rl_timing_count.restype = C.c_uint
alg, S, A, B, kw = bench.WORKLOADS['vlsac_halfcheetah_f256_b256']
torch.manual_seed(0)
agent = bench.make_agent(alg, S, A, B, kw)
buf, _ = bench.synth_buffer(S, A, 0)
for _ in range(20): agent.train(buf, B)
torch.cuda.synchronize()
NL = 80; CAP = NL * 2048
tb = torch.zeros(CAP * 12, dtype=torch.int64, device='cuda')          # RlTimRec is 96 bytes
assert raw.rl_timing_buffer(C.c_void_p(tb.data_ptr()), CAP) == 0
for _ in range(3): agent.train(buf, B)
torch.cuda.synchronize()
tb.zero_(); torch.cuda.synchronize()
assert raw.rl_timing_buffer(C.c_void_p(tb.data_ptr()), CAP) == 0     # reset the launch counter
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); agent.train(buf, B); e1.record()
torch.cuda.synchronize()
print('this train(): %.1f us by events' % (e0.elapsed_time(e1) * 1e3))
nl = raw.rl_timing_count()
rec = tb.cpu().numpy().reshape(CAP, 12)
print(f'{nl} gemm16 launches; wall times in us (100 MHz clock); stamps in shader cycles AFTER ENTRY (median over the sampled workgroups; / 2200 = us)')
print('fast = front end that loads from preloaded scalars (gemm16_fast*_kernel): there "loads" (first operand loads issued) precedes "record"; in the record form "loads" = slot loads about to issue')
print(' #   WGs fast  gap_prev  span  |  task-known   record    loads  slots-issued  mfma-done  reduced    exit  |  life us (max)')
prev = None; tot = 0.0
for k in range(nl):
    r = rec[k * 2048:(k + 1) * 2048]; r = r[(r[:, 11] >> 32) == 1]
    if not len(r): continue
    w0, w4 = r[:, 0], r[:, 1]
    ent = r[:, 2]
    off = lambda col: np.median(r[:, col] - ent)           # c[q] sits in column 2 + q
    grid = int(r[0, 10] & 0xffffffff)
    fast = int(np.median(r[:, 7] - r[:, 3]) < 0)          # first operand loads stamped BEFORE the record was in registers
    gap = (w0.min() - prev) / 100 if prev is not None else 0.0
    life = (r[:, 6] - ent) / 2200.0
    print(f"{k:3d} {grid:5d}  {fast:3d}  {gap:7.2f} {(w4.max()-w0.min())/100:6.2f}  |  {off(9):9.0f} {off(3):8.0f} {off(7):8.0f} {off(8):12.0f} {off(4):10.0f} {off(5):8.0f} {off(6):7.0f}  | {np.median(life):6.2f} ({life.max():.2f})")
    tot += (w4.max() - w0.min()) / 100
    prev = w4.max()
print('sum of gemm16 kernel spans %.1f us' % tot)
