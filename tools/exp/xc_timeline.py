"""In-kernel timeline of the persistent chain launches (csrc/xchain.hip) of one graph-replayed vlsac train().
Needs the instrumented build:
   OBJDIR=.obj_xct OUTNAME=librlrep_hip_xct.so EXTRA_FLAGS="-DRL_TIMING -DRL_TIMING_XC" bash rlrep_amd/csrc/build.sh
   RLREP_LIB=$PWD/rlrep_amd/lib/librlrep_hip_xct.so RLREP_PIPELINE=0 python tools/exp/xc_timeline.py
Per phase and workgroup: 100 MHz wall clock at phase entry / wait passed / tiles done / flag published (thread 0), and the shader clock
inside the first tile (group 0's workgroups only: record in registers, MFMAs issued, reduction barrier passed, tile done)."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
import numpy as np, torch
import bench
from rlrep_amd import _lib

raw = C.CDLL(_lib.LIB_PATH)
raw.rl_xc_timing_buffer.argtypes = [C.c_void_p, C.c_uint]; raw.rl_xc_timing_buffer.restype = C.c_int
raw.rl_xc_timing_count.restype = C.c_uint
alg, S, A, B, kw = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else 'vlsac_halfcheetah_f256_b256']
torch.manual_seed(0)
agent = bench.make_agent(alg, S, A, B, kw)
buf, _ = bench.synth_buffer(S, A, 0)
for _ in range(30): agent.train(buf, B)
torch.cuda.synchronize()
NL = 16; CAP = NL * 32 * 512 * 12
tb = torch.zeros(CAP, dtype=torch.int64, device='cuda')
for _ in range(3): agent.train(buf, B)
torch.cuda.synchronize()
assert raw.rl_xc_timing_buffer(C.c_void_p(tb.data_ptr()), CAP) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); agent.train(buf, B); e1.record()
agent.flush(); torch.cuda.synchronize()
print('this train(): %.1f us by events' % (e0.elapsed_time(e1) * 1e3))
nl = raw.rl_xc_timing_count()
rec = tb.cpu().numpy().reshape(NL, 32, 512, 12)
kinds = {0: 'gemm', 1: 'heads_vae', 2: 'qhead_c', 3: 'qhead_a'}
for k in range(min(nl, NL)):
    r = rec[k]
    live = r[:, :, 0] != 0
    nph = int(live.any(axis=1).sum())
    nblk = int(live[0].sum())
    t_first = r[0][live[0]][:, 0].min()
    t_last = r[nph - 1][live[nph - 1]][:, 3].max()
    print(f'launch {k}: {nph} phases, {nblk} workgroups, in-kernel span {(t_last - t_first) / 100:.2f} us')
    print('  ph kind      tiles/grp | phase span | median per WG: wait  tiles  publish | slowest WG tiles | first tile (group 0, cycles): record  slots  ev-issue  loads+mfma  reduce  epilogue')
    prev_end = t_first
    for p in range(nph):
        x = r[p][live[p]]
        w0, w1, w2, w3 = x[:, 0], x[:, 1], x[:, 2], x[:, 3]
        kind = int(x[0, 9] >> 32); tiles = int(x[0, 9] & 0xffffffff)
        c = np.stack([x[:, 4], x[:, 5], x[:, 10], x[:, 11], x[:, 6], x[:, 7], x[:, 8]], axis=1); cc = c[(c[:, 0] != 0) & (c[:, 6] != 0) & (c[:, 1] != 0)]
        cyc = np.median(np.diff(cc, axis=1), axis=0) if len(cc) else np.zeros(6)
        end = w3.max()
        print(f'  {p:2d} {kinds.get(kind, "?"):10s} {tiles:6d}   | {(end - prev_end) / 100:8.2f}   | {np.median(w1 - w0) / 100:14.2f} {np.median(w2 - w1) / 100:6.2f} {np.median(w3 - w2) / 100:7.2f}  | {(w2 - w1).max() / 100:10.2f}       | {cyc[0]:8.0f} {cyc[1]:6.0f} {cyc[2]:8.0f} {cyc[3]:10.0f} {cyc[4]:8.0f} {cyc[5]:8.0f}')
        prev_end = end
