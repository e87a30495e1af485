"""Soak: N pipelined train() calls against N sequential ones (same seeds, device Philox) -- final parameters must agree."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
import numpy as np, torch
import bench
from fixture_io import rel_l2
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
wl = sys.argv[1] if len(sys.argv) > 1 else 'vlsac_halfcheetah_f256_b256'
alg, S, A, B, kw = bench.WORKLOADS[wl]
outs = []
for pipe in (True, False):
    torch.manual_seed(0)
    import importlib
    name = {'sac': 'SACAgent', 'vlsac': 'VLSACAgent', 'ctrlsac': 'CTRLSACAgent', 'spedersac': 'SPEDERSACAgent'}[alg]
    cls = getattr(importlib.import_module(f'rlrep_amd.agent.{alg}.{alg}_agent'), name)
    agent = cls(state_dim=S, action_dim=A, action_space=bench.Space(A), max_batch=B, pipeline=pipe, seed=777, **kw)
    buf, _ = bench.synth_buffer(S, A, 0)
    for i in range(N):
        agent.train(buf, B)
        if i % 501 == 500:
            agent.select_action(np.zeros(S, np.float32))
    st = {k: v.numpy().copy() for k, v in agent.core.state().items()}
    outs.append(st)
worst = max((rel_l2(outs[0][k], outs[1][k]), k) for k in outs[1])
finite = all(np.all(np.isfinite(v)) for v in outs[0].values())
print(f'{wl}: {N} train() calls, worst relative L2 difference pipelined vs sequential: {worst[0]:.3e} ({worst[1]}); finite={finite}')
