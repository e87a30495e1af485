#!/usr/bin/env python3
"""The 256 x 128 persistent bf16x3 tile (csrc/gemm_x3w.h) through rlrep_gemm(engine 2, bt 256): parity on small / ragged / split-K shapes against
float64, then the three 202-GFLOP products of diffsrsac's nabla-mu head timed against the 128 x 128 tile (bt 128).
    python tools/exp/x3w_check.py [quick]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
from test_gemm_engines import run_gemm, rel
from rlrep_amd import _lib

bad = 0
for mode in ('fwd', 'dx', 'dw'):
    cases = [dict(R=256, Cn=128, K=64, splits=1), dict(R=512, Cn=256, K=128, splits=1), dict(R=148, Cn=92, K=100, splits=1), dict(R=1032, Cn=644, K=196, splits=1),
             dict(R=300, Cn=132, K=1024, splits=3, accum=(mode != 'fwd')), dict(R=768, Cn=640, K=512, act='elu' if mode != 'dw' else 'none'),
             dict(R=2048, Cn=1024, K=96, splits=1), dict(R=260, Cn=4100, K=40, splits=1), dict(R=5000, Cn=128, K=320, splits=2)]
    for i, c in enumerate(cases):
        kw = dict(c); R, Cn, K = kw.pop('R'), kw.pop('Cn'), kw.pop('K')
        got, want, extra = run_gemm(2, mode, R, Cn, K, bt=256, seed=100 + i, **kw)
        e = rel(got, want); e2 = rel(extra[0], extra[1]) if extra is not None else 0.0
        ok = np.all(np.isfinite(got)) and e < 1e-5 and e2 < 1e-5
        bad += 0 if ok else 1
        print(f'{mode} {R}x{Cn}x{K} {kw}: rel {e:.2e} second {e2:.2e} {"ok" if ok else "FAIL"}', flush=True)
print('parity failures:', bad, flush=True)
if bad or (len(sys.argv) > 1 and sys.argv[1] == 'parity'):
    sys.exit(1 if bad else 0)


def timeit(mode, R, Cn, K, bt, reps=5, splits=0):
    la, lb = {'fwd': (0, 0), 'dx': (0, 1), 'dw': (1, 1)}[mode]
    A = torch.randn((K, R) if la else (R, K), device='cuda')
    B = torch.randn((K, Cn) if lb else (Cn, K), device='cuda')
    C = torch.empty(R, Cn, device='cuda')
    ws = torch.empty(min(32 * R * (Cn + 5), 40_000_000 + 2 * (R + 128) * (Cn + 133)), device='cuda')
    epi = {'fwd': 0, 'dx': 1, 'dw': 3}[mode]
    st = torch.cuda.current_stream().cuda_stream

    def go():
        _lib.check(_lib.lib.rlrep_gemm(2, la, lb, A.data_ptr(), A.shape[1], B.data_ptr(), B.shape[1], C.data_ptr(), Cn, R, Cn, K, epi, 0, 0, None, None, Cn, None,
                                       bt, splits, ws.data_ptr(), ws.numel(), st), 'gemm')
    for _ in range(2):
        go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        go()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for name, mode, R, Cn, K in (('nabla-mu fwd', 'fwd', 2048, 96256, 512), ('nabla-mu dX', 'dx', 2048, 512, 96256), ('nabla-mu dW', 'dw', 96256, 512, 2048),
                             ('square 4096 fwd', 'fwd', 4096, 4096, 4096), ('square 4096 dx', 'dx', 4096, 4096, 4096), ('square 4096 dW', 'dw', 4096, 4096, 4096)):
    fl = 2.0 * R * Cn * K
    line = f'{name:18s} {R:6d} x {Cn:6d} x {K:6d}'
    for bt in (128, 256):
        for sp in ((0,) if mode != 'dx' or K < 50000 else (0, 8, 16)):
            us = timeit(mode, R, Cn, K, bt, splits=sp)
            line += f' | bt {bt} splits {sp}: {us:8.1f} us {fl / us / 1e6:6.1f} TF'
    print(line, flush=True)
