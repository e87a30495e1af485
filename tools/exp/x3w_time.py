#!/usr/bin/env python3
"""Timing only: rlrep_gemm(engine 2, bt = BT env, default 256) on the nabla-mu head's three products and 4096^3 (RLREP_LIB selects an ablation build)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
from rlrep_amd import _lib
BT = int(os.environ.get('BT', 256))


def timeit(mode, R, Cn, K, bt, reps=5, splits=0):
    la, lb = {'fwd': (0, 0), 'dx': (0, 1), 'dw': (1, 1)}[mode]
    mk = {'randn': torch.randn, 'zeros': torch.zeros, 'ones': torch.ones}[os.environ.get('DATA', 'randn')]          # (operand values decide the matrix pipe's power, and with it the clock)
    A = mk((K, R) if la else (R, K), device='cuda')
    B = mk((K, Cn) if lb else (Cn, K), device='cuda')
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


timeit('fwd', 4096, 4096, 4096, BT, reps=30)          # (clocks up before the first measured product)
line = os.path.basename(os.environ.get('RLREP_LIB', 'product')) + f' bt {BT} {os.environ.get("DATA", "randn")}:'
for name, mode, R, Cn, K, sp in (('fwd', 'fwd', 2048, 96256, 512, 0), ('dX', 'dx', 2048, 512, 96256, 16), ('dW', 'dw', 96256, 512, 2048, 0), ('4096 fwd', 'fwd', 4096, 4096, 4096, 0),
                                 ('4096 dW', 'dw', 4096, 4096, 4096, 0)):
    us = timeit(mode, R, Cn, K, BT, splits=sp)
    line += f' | {name} {us:7.1f} us {2.0 * R * Cn * K / us / 1e6:6.1f} TF'
print(line, flush=True)
