import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0]=[ROOT, ROOT+'/tests', ROOT+'/tests/golden']
import test_large_dims as T
for (S,A,B,F,H) in [(5,2,5,64,32), (7,3,13,128,64), (6,2,8,64,96), (9,4,260,64,32), (3,1,31,96,160)]:
    T._run('vlsac', ('rlrep_amd.agent.vlsac.vlsac_agent','VLSACAgent'), S, A, B, dict(hidden_dim=H, feature_dim=F, extra_feature_steps=0), trains=2)
print('fuzz ok')
