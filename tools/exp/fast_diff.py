"""Which tensors differ between the fast front end and RLREP_DISABLE=gemm16_fast after n train() calls (two processes, same seeds)."""
import os, sys, subprocess, pickle
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
    import numpy as np, torch, synth
    from rlrep_amd.utils.buffer import ReplayBuffer
    from rlrep_amd.agent.vlsac.vlsac_agent import VLSACAgent
    class Sp:
        low, high = -np.ones(6, np.float32), np.ones(6, np.float32)
    data = synth.replay(17, 6, 8192, seed=0)
    torch.manual_seed(0)
    kw = dict(graph=False) if sys.argv[3] == 'eager' else dict(pipeline=True)
    agent = VLSACAgent(state_dim=17, action_dim=6, action_space=Sp(), max_batch=256, seed=78, hidden_dim=256, feature_dim=256, extra_feature_steps=3, **kw)
    buf = ReplayBuffer(17, 6, max_size=8192)
    buf.load(data['state'], data['action'], data['next_state'], data['reward'], data['done'])
    for _ in range(int(sys.argv[2])):
        agent.train(buf, 256)
    agent.flush(); torch.cuda.synchronize()
    st = {k: v.numpy().copy() for k, v in agent.core.state().items()}
    pickle.dump(st, open(sys.argv[4], 'wb'))
    sys.exit(0)
n, mode = sys.argv[1], sys.argv[2]
outs = []
for arm, env in (('fast', {}), ('nofast', {'RLREP_DISABLE': 'gemm16_fast'})):
    f = f'/tmp/fast_diff_{arm}.pkl'
    subprocess.check_call([sys.executable, __file__, 'child', n, mode, f], env=dict(os.environ, **env))
    outs.append(pickle.load(open(f, 'rb')))
import numpy as np
for k in outs[0]:
    a, b = outs[0][k], outs[1][k]
    if not np.array_equal(a, b):
        print(k, 'differs: max abs', float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)))), 'of', float(np.max(np.abs(b))))
print('done')
