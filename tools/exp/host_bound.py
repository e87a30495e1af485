"""Is a workload's train() loop bound by the HOST (python + graph launch) or by the device chain?  Times n calls without synchronising,
then the drain: a drain near zero says the device kept up with the host, i.e. the host sets the rate.
    python tools/exp/host_bound.py sac_halfcheetah_b256"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
import torch, bench
for name in sys.argv[1:] or ['sac_halfcheetah_b256']:
    alg, S, A, B, kw = bench.WORKLOADS[name]
    torch.manual_seed(0)
    agent = bench.make_agent(alg, S, A, B, kw)
    buf, _ = bench.synth_buffer(S, A, 0)
    for _ in range(300): agent.train(buf, B)
    agent.flush(); torch.cuda.synchronize()
    for rep in range(3):
        n = 2000
        t0 = time.perf_counter()
        for _ in range(n): agent.train(buf, B)
        t1 = time.perf_counter(); agent.flush(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f'{name}: host {1e6 * (t1 - t0) / n:.1f} us per call, drain {1e3 * (t2 - t1):.2f} ms, total {1e6 * (t2 - t0) / n:.1f} us per call')
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(2000): agent.train(buf, B)
    pr.disable(); agent.flush(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
