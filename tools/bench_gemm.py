#!/usr/bin/env python3
"""Time the two GEMM engines (rlrep_gemm) on the shapes the step programs route to gemm_lds.hip.
    python tools/bench_gemm.py            (on the GPU box)
Prints one line per shape/engine: microseconds per launch and TFLOP/s against the 157.3 TF fp32-MFMA peak."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlrep_amd import _lib

SHAPES = [
    # name, mode, R, Cn, K
    ('ctrlsac phi.l2 fwd', 'fwd', 256, 1024, 1024),
    ('ctrlsac phi.l3 fwd', 'fwd', 256, 2048, 1024),
    ('ctrlsac critic l1|l4 fwd', 'fwd', 256, 2048, 2048),
    ('ctrlsac critic l1|l4 dx', 'dx', 256, 2048, 2048),
    ('ctrlsac score matrix', 'fwd', 256, 256, 2048),
    ('ctrlsac phi.l3 dx', 'dx', 256, 1024, 2048),
    ('ctrlsac phi.l2 dW', 'dw', 1024, 1024, 256),
    ('ctrlsac phi.l3 dW', 'dw', 2048, 1024, 256),
    ('spedersac phi fwd', 'fwd', 2048, 512, 512),
    ('spedersac phi dx', 'dx', 2048, 512, 512),
    ('spedersac phi dW', 'dw', 512, 512, 2048),
    ('spedersac critic l1|l4 fwd', 'fwd', 1024, 512, 512),
    ('spedersac critic l1|l4 dW', 'dw', 512, 512, 1024),
    ('spedersac critic l2 fwd', 'fwd', 1024, 256, 256),
    ('diffsr nabla-mu head fwd', 'fwd', 2048, 96256, 512),
    ('diffsr nabla-mu head dx', 'dx', 2048, 512, 96256),
    ('diffsr nabla-mu head dW', 'dw', 96256, 512, 2048),
    ('diffsr head fwd with W^T (k-major B)', 'dx', 2048, 96256, 512),
    ('square 4096 dx', 'dx', 4096, 4096, 4096),
    ('square 4096 dW', 'dw', 4096, 4096, 4096),
    ('square 4096', 'fwd', 4096, 4096, 4096),
]


def run(engine, mode, R, Cn, K, reps, bt=int(os.environ.get('BT', 0)), splits=int(os.environ.get('SPLITS', 0))):
    la, lb = {'fwd': (0, 0), 'dx': (0, 1), 'dw': (1, 1)}[mode]
    A = torch.randn((K, R) if la else (R, K), device='cuda')
    B = torch.randn((K, Cn) if lb else (Cn, K), device='cuda')
    C = torch.empty(R, Cn, device='cuda')
    ws = torch.empty(min(32 * R * (Cn + 1), 10_000_000 + 2 * (R + 128) * (Cn + 129)), device='cuda')
    epi = {'fwd': 0, 'dx': 1, 'dw': 3}[mode]
    st = torch.cuda.current_stream().cuda_stream

    def go():
        rc = _lib.lib.rlrep_gemm(engine, la, lb, A.data_ptr(), A.shape[1], B.data_ptr(), B.shape[1], C.data_ptr(), Cn, R, Cn, K,
                                 epi, 0, 0, None, None, Cn, None, bt, splits, ws.data_ptr(), ws.numel(), st)
        _lib.check(rc, 'gemm')
    for _ in range(3):
        go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        go()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


if __name__ == '__main__':
    only = os.environ.get('ONLY')          # substring of a shape name
    engines = [int(x) for x in os.environ.get('ENGINES', '0,1,2').split(',')]
    for name, mode, R, Cn, K in SHAPES:
        if only and only not in name:
            continue
        fl = 2.0 * R * Cn * K
        reps = 5 if fl > 5e10 else 50
        line = f'{name:28s} {mode:3s} {R:6d} x {Cn:6d} x {K:6d}  {fl / 1e9:8.2f} GF'
        for eng, label in ((0, 'gemm16'), (1, 'gemm_lds'), (2, 'bf16x3')):
            if eng not in engines:
                continue
            if (eng == 0 and fl > 3e10 and os.environ.get('SKIP_SLOW')) or (eng == 2 and fl < float(os.environ.get('X3_MIN_FLOP', 1e9))):
                continue
            us = run(eng, mode, R, Cn, K, reps)
            line += f' | {label} {us:9.1f} us {fl / us / 1e6:6.1f} TF'
        print(line, flush=True)
