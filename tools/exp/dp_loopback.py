"""What the in-launch data-parallel exchange COSTS, measured without a second GPU and without process time-slicing.

N replicas of one agent inside ONE process on ONE GPU (rlrep_amd/comm.py LoopbackGroup: every replica's gradient block is plain device memory,
the peers are plain pointers), each on its own pair of streams, each driven by its own host thread:

    attached     the replicas are data-parallel ranks: every optimizer launch waits for the peers' gradients, sums all ranks' arenas in rank
                 order (one-shot, or two-shot for the large slices when N >= 3) and runs the DONE handshake -- adam_dp_kernel<1 | 2>; spedersac /
                 ctrlsac also carry their batch-coupled exchanges inside the step program
    unattached   the SAME N agents as independent single-GPU agents (world_size = 1, adam_kernel, no exchange): same chip load, no protocol

Both arms run the default train() (hipGraph replay, two chains per replica) for --calls calls per replica; aggregate train()/s = N * calls / wall.
attached / unattached is the protocol's price when the wire is free (same HBM): flags, system-scope loads, the second read of the gradients,
the handshakes' round trips.  Run under `rocprofv3 --kernel-trace --stats` the two arms' optimizer kernels can be read side by side
(adam_dp_kernel<1>, <2> vs adam_kernel).

    python tools/exp/dp_loopback.py --workload vlsac_halfcheetah_f256_b256 --world 2 --calls 400
"""
import argparse
import json
import os
import sys
import threading
import time

os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]

import numpy as np   # noqa: E402
import torch         # noqa: E402


def run_arm(workload, world, calls, warm, attached, two_shot_kb):
    import bench
    from rlrep_amd.comm import LoopbackGroup
    alg, S, A, B, kw = bench.WORKLOADS[workload]
    if two_shot_kb is not None:
        os.environ['RLREP_ENABLE'] = f'dp_two_shot_kb={two_shot_kb}'
    grp = LoopbackGroup(world) if attached else None
    if grp is not None:
        grp.timeout_s = 60.0
    agents, bufs = [], []
    for r in range(world):
        torch.manual_seed(0)
        extra = dict(loopback=(grp, r)) if attached else {}
        agents.append(bench.make_agent(alg, S, A, B, {**kw, **extra}))
        bufs.append(bench.synth_buffer(S, A, seed=r)[0])
    if not attached:
        # independent replicas still need their own stream pairs: hand each one pair `r` of the mutually concurrent set
        import rlrep_amd.agent.sac.sac_agent as sa
        for r, a in enumerate(agents):
            a._loopback_index = r
        orig = sa._concurrent_stream_pair

        def pair(core, index=0):
            for r, a in enumerate(agents):
                if a.core is core:
                    return orig(core, r)
            return orig(core, index)
        sa._concurrent_stream_pair = pair
    streams = [torch.cuda.Stream() for _ in range(world)]
    for r in range(world):                      # graphs are captured one replica after the other (a capture synchronises the device), replayed side by side
        with torch.cuda.stream(streams[r]):
            agents[r].prepare(bufs[r], B)
    torch.cuda.synchronize()
    barrier = threading.Barrier(world + 1)
    errs = []

    def body(r):
        try:
            with torch.cuda.stream(streams[r]):
                for _ in range(warm):
                    agents[r].train(bufs[r], B)
                agents[r].flush()
                streams[r].synchronize()
                barrier.wait()
                barrier.wait()
                for _ in range(calls):
                    agents[r].train(bufs[r], B)
                agents[r].flush()
                streams[r].synchronize()
                barrier.wait()
        except BaseException as e:          # noqa: BLE001
            errs.append((r, repr(e)))
            barrier.abort()

    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    barrier.wait()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    barrier.wait()
    barrier.wait()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for t in ts:
        t.join()
    if errs:
        raise RuntimeError(str(errs))
    out = dict(arm='attached' if attached else 'unattached', world=world, calls_each=calls, wall_s=round(dt, 4),
               aggregate_train_per_s=round(world * calls / dt, 1), per_replica_train_per_s=round(calls / dt, 1),
               form=('pipe' if agents[0]._pipe is not None else 'graph'))
    if attached:
        out['fused_groups'] = sorted(agents[0].core.fused_groups)
        out['feature_exchanges_left_outside'] = int(agents[0].core.feature_exchange_count())
        out['status'] = [int(a.core.exchange.status(raise_on_error=False)) for a in agents]
        chk = [(int(a.core.params.view(torch.int32).to(torch.int64).sum()), int(a.core.targets.view(torch.int32).to(torch.int64).sum())) for a in agents]
        out['replicas_identical'] = all(c == chk[0] for c in chk)
    del agents
    if grp is not None:
        grp.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='vlsac_halfcheetah_f256_b256')
    ap.add_argument('--world', type=int, default=2)
    ap.add_argument('--calls', type=int, default=400)
    ap.add_argument('--warm', type=int, default=120)
    ap.add_argument('--arms', default='unattached,attached')
    ap.add_argument('--two-shot-kb', default=None, help='slices of at least this many KB take the two-shot form (world >= 3); 0: never')
    args = ap.parse_args()
    torch.cuda.set_device(0)
    res = []
    for arm in args.arms.split(','):
        res.append(run_arm(args.workload, args.world, args.calls, args.warm, arm == 'attached', args.two_shot_kb))
        print(json.dumps(res[-1]), flush=True)
    by = {r['arm']: r for r in res}
    if 'attached' in by and 'unattached' in by:
        print(json.dumps(dict(workload=args.workload, world=args.world, two_shot_kb=args.two_shot_kb,
                              attached_over_unattached=round(by['attached']['aggregate_train_per_s'] / by['unattached']['aggregate_train_per_s'], 4))), flush=True)


if __name__ == '__main__':
    main()
