import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
import numpy as np, torch
import synth
from rlrep_amd.utils.buffer import ReplayBuffer
from rlrep_amd.agent.vlsac.vlsac_agent import VLSACAgent
class Sp:
    low = -np.ones(6, np.float32); high = np.ones(6, np.float32)
data = synth.replay(17, 6, 8192, seed=0)
outs = []
for pipe in (True, False):
    torch.manual_seed(0)
    agent = VLSACAgent(state_dim=17, action_dim=6, action_space=Sp(), max_batch=256, pipeline=pipe, seed=99, hidden_dim=256, feature_dim=256, extra_feature_steps=3)
    buf = ReplayBuffer(17, 6, max_size=8192)
    buf.load(data['state'], data['action'], data['next_state'], data['reward'], data['done'])
    n = int(os.environ.get('NTRAIN', '1'))
    for i in range(n): agent.train(buf, 256)
    outs.append({k: v.numpy().copy() for k, v in agent.core.state().items()})
for k in outs[0]:
    a, b = outs[0][k].astype(np.float64), outs[1][k].astype(np.float64)
    d = np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
    if d > 0: print(f'{k}: rel l2 {d:.3e}  max abs {np.abs(a - b).max():.3e}')
print('done')
