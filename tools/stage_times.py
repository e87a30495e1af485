import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
import numpy as np, torch
import bench
alg,S,A,B,kw = bench.WORKLOADS[sys.argv[1] if len(sys.argv)>1 else 'vlsac_halfcheetah_f256_b256']
torch.manual_seed(0)
agent = bench.make_agent(alg,S,A,B,kw)
buf,_ = bench.synth_buffer(S,A,0)
for _ in range(20): agent.train(buf,B)
torch.cuda.synchronize()
# whole-train GPU time via events
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): agent.train(buf,B)
e1.record(); torch.cuda.synchronize()
print('train() GPU time per call (events, graph=%s): %.1f us'%(agent.use_graph, e0.elapsed_time(e1)*1e3/200))
core=agent.core
WARM=int(os.environ.get('STAGE_WARM','20'))      # replays of the 50-launch graph before timing (clock ramp)
ONLY=os.environ.get('STAGE_ONLY')               # substring filter on the stage name
names={0:'feature_bwd',1:'feature_apply',2:'critic_bwd',3:'critic_apply',4:'actor_bwd',5:'actor_apply',6:'update_target'}
tot_all=0
for p in range(7):
    st=core.stages(p); tot=0
    for i,n in enumerate(st):
        if ONLY and ONLY not in n: continue
        for _ in range(10): core.run_stage(p,i)
        torch.cuda.synchronize()
        g=torch.cuda.CUDAGraph(); s=torch.cuda.Stream()
        with torch.cuda.graph(g, stream=s):
            for _ in range(50): core.run_stage(p,i)
        for _ in range(WARM): g.replay()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(8): g.replay()
        e1.record(); torch.cuda.synchronize()
        us=e0.elapsed_time(e1)*1e3/400
        tot+=us
        print(f'  {names[p]:14s} {i:2d} {us:7.2f} us  {n}')
    mult = 4 if p<2 else 1
    print(f'{names[p]}: {tot:.1f} us x{mult}')
    tot_all+=tot*mult
print('sum of stage times per train():', tot_all)
