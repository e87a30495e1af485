"""Stage names of an agent's step programs: python tools/exp/list_stages.py <workload>"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
import torch, bench
name = sys.argv[1] if len(sys.argv) > 1 else 'sac_halfcheetah_b256'
alg, S, A, B, kw = bench.WORKLOADS[name]
agent = bench.make_agent(alg, S, A, B, kw)
buf, _ = bench.synth_buffer(S, A, 0)
for _ in range(3): agent.train(buf, B)
agent.flush()
names = {0: 'feat_bwd', 1: 'feat_apply', 2: 'critic_bwd', 3: 'critic_apply', 4: 'actor_bwd', 5: 'actor_apply', 6: 'upd_target', 7: 'critic_bwd_h', 8: 'feat_bwd_h', 9: 'critic_bwd_h2'}
for p, n in names.items():
    st = agent.core.stages(p)
    print(f'{n} ({len(st)}):')
    for s in st: print('    ', s)
print('launches per train():', getattr(agent, '_graph_launches', None), (agent._pipe or {}).get('launches') if getattr(agent, '_pipe', None) else None)
