"""Per-op timeline of the row-program launch of a vlsac feature step (rowprog.hip diagnostic stamps: 100 MHz wall clock at the start of
every op, thread 0 of every workgroup).  Prints, per program, the median over row blocks of each op's duration and the launch span."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
import numpy as np, torch
import bench
from rlrep_amd._lib import lib
alg, S, A, B, kw = bench.WORKLOADS['vlsac_halfcheetah_f256_b256']
torch.manual_seed(0)
agent = bench.make_agent(alg, S, A, B, kw)
buf, _ = bench.synth_buffer(S, A, 0)
for _ in range(5): agent.train(buf, B)
agent.flush(); torch.cuda.synchronize()
core = agent.core
prog = int(sys.argv[1]) if len(sys.argv) > 1 else 0
stage = [i for i, n in enumerate(core.stages(prog)) if n.startswith('row programs')][0]
nblk = 256
tim = torch.zeros(nblk * 512, dtype=torch.int64, device='cuda')
lib.rl_rowprog_timing.argtypes = [C.c_void_p]; lib.rl_rowprog_timing.restype = C.c_int
for _ in range(20): core.run_stage(prog, stage)
torch.cuda.synchronize()
assert lib.rl_rowprog_timing(C.c_void_p(tim.data_ptr())) == 0
core.run_stage(prog, stage); torch.cuda.synchronize()
lib.rl_rowprog_timing(C.c_void_p(0))
tt = tim.cpu().numpy().reshape(nblk, 8, 64)
t, cyc, tk, tb = tt[:, 0], tt[:, 1], tt[:, 2], tt[:, 3]
tv, tl, tg, tf = tt[:, 4], tt[:, 5], tt[:, 6], tt[:, 7]
used = [b for b in range(nblk) if t[b, 0] != 0]
t0 = min(t[b, 0] for b in used)
nrb = (B + 15) // 16
print(f'{len(used)} workgroups; launch span {(max(t[b].max() for b in used) - t0) / 100:.1f} us')
for p in range(len(used) // nrb):
    rows = [t[b] for b in used[p * nrb:(p + 1) * nrb]]
    crow = cyc[used[p * nrb]]
    nops = int(np.count_nonzero(rows[0])) - 1
    print(f'program {p}: {nops} ops; start (median) {np.median([r[0] - t0 for r in rows]) / 100:.2f} us, end {np.median([r[nops] - t0 for r in rows]) / 100:.2f} us')
    for oi in range(nops):
        d = [(r[oi + 1] - r[oi]) / 100 for r in rows]
        mhz = (crow[oi + 1] - crow[oi]) / max((rows[0][oi + 1] - rows[0][oi]) / 100, 1e-9)
        print(f'   op {oi:2d}: median {np.median(d):6.2f} us  min {min(d):6.2f}  max {max(d):6.2f}   (starts at {np.median([r[oi] - t0 for r in rows]) / 100:6.2f})  shader clock {mhz:6.0f} MHz' + (f'   [k-loop {(tk[used[p * nrb]][oi] - rows[0][oi]) / 100:5.2f}  barrier {(tb[used[p * nrb]][oi] - tk[used[p * nrb]][oi]) / 100:5.2f}  values {(tv[used[p * nrb]][oi] - tb[used[p * nrb]][oi]) / 100:5.2f} lds {(tl[used[p * nrb]][oi] - tv[used[p * nrb]][oi]) / 100:5.2f} gstores {(tg[used[p * nrb]][oi] - tl[used[p * nrb]][oi]) / 100:5.2f} endbar {(tf[used[p * nrb]][oi] - tg[used[p * nrb]][oi]) / 100:5.2f} next {(rows[0][oi + 1] - tf[used[p * nrb]][oi]) / 100:5.2f}]' if tk[used[p * nrb]][oi] else ''))
