#!/usr/bin/env python3
"""Phase timeline of nc_fwd_kernel (instrumented build: EXTRA_FLAGS=-DRL_TIMING_NC bash rlrep_amd/csrc/build.sh into a
separate library, RLREP_LIB=<that .so>): per-workgroup shader-clock stamps at entry / after table staging / after the MFMA
loop / exit, and wall-clock (100 MHz) entry / exit, for the critic-step launch (4 heads)."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
import bench
from rlrep_amd import _lib
alg, S, A, B, kw = bench.WORKLOADS['vlsac_halfcheetah_f256_b256']
agent = bench.make_agent(alg, S, A, B, kw)
buf, _ = bench.synth_buffer(S, A, 0)
for _ in range(5):
    agent.train(buf, B)
core = agent.core
PROG = int(os.environ.get('NCT_PROG', '2')); PREFIX = os.environ.get('NCT_STAGE', 'noise critic l1/l4')     # e.g. NCT_PROG=4 NCT_STAGE='noise critic dX'
names = core.stages(PROG)
s = [i for i, n in enumerate(names) if n.startswith(PREFIX)][0]
for _ in range(5):
    core.run_stage(PROG, s)
torch.cuda.synchronize()
if os.environ.get('NCT_GRAPH'):
    # the stamps of the LAST of 50 back-to-back launches inside a graph (the condition tools/stage_times.py times)
    g = torch.cuda.CUDAGraph(); st = torch.cuda.Stream()
    with torch.cuda.graph(g, stream=st):
        for _ in range(50):
            core.run_stage(PROG, s)
    for _ in range(int(os.environ.get('NCT_WARM', '200'))):       # ~0.2 s of the same launch: clocks at their sustained level
        g.replay()
    torch.cuda.synchronize()
n = 6 * 4096
host = (C.c_ulonglong * n)()
_lib.lib.rl_nc_timing_fetch.argtypes = [C.c_void_p, C.c_int]
assert _lib.lib.rl_nc_timing_fetch(host, n) == 0
t = np.array(host, dtype=np.float64).reshape(4096, 6)
nwg = int((t[:, 0] > 0).sum())
t = t[:nwg]
w0 = t[:, 4].min()
print('workgroups', nwg)
print('launch wall span (first entry -> last exit): %.2f us' % ((t[:, 5].max() - w0) / 100.0))
print('entry spread: %.2f us' % ((t[:, 4].max() - w0) / 100.0))
for name, a, b in (('staging', 0, 1), ('mfma loop', 1, 2), ('epilogue', 2, 3), ('whole', 0, 3)):
    d = t[:, b] - t[:, a]
    print('%-10s cycles: median %8.0f  p10 %8.0f  p90 %8.0f' % (name, np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
dur = (t[:, 5] - t[:, 4]) / 100.0
print('per-workgroup wall duration: median %.2f us, min %.2f, max %.2f; implied clock %.0f MHz' % (np.median(dur), dur.min(), dur.max(), np.median((t[:, 3] - t[:, 0]) / dur)))
