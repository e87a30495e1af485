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
    from rlrep_amd.comm import concurrent_streams
    ss = concurrent_streams(2 * world)          # (HIP gives a process four concurrent hardware queues: world <= 2 for the two-chain form)
    if not attached:
        for r, a in enumerate(agents):          # independent replicas: the same disjoint stream pairs the attached ones take
            a._stream_pair_override = (ss[2 * r], ss[2 * r + 1])
    streams = [ss[2 * r] for r in range(world)]      # the caller's stream of replica r = one of ITS OWN two (a fifth stream would share a hardware queue with some chain)
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


def run_alone(workload, calls, warm, attached):
    """ONE replica on the chip: a plain single-GPU agent, or rank 0 of a two-rank group whose peer is marked as always arrived
    (rlrep_comm_debug_preset: every wait passes at once, the peer's arena holds zeros) -- the protocol's own price (flag reads, the handshake's
    stores and polls, system-scope loads of two arenas where one gradient was read) with no waiting and no contention in it."""
    import bench
    from rlrep_amd._lib import lib, check
    from rlrep_amd.comm import LoopbackGroup
    alg, S, A, B, kw = bench.WORKLOADS[workload]
    grp = LoopbackGroup(2) if attached else None
    torch.manual_seed(0)
    agent = bench.make_agent(alg, S, A, B, {**kw, **(dict(loopback=(grp, 0)) if attached else {})})
    buf = bench.synth_buffer(S, A, seed=0)[0]
    if attached:
        for ch in range(8):
            check(lib.rlrep_comm_debug_preset(grp[0].h, ch, 1 << 30), 'debug_preset')
    for _ in range(warm):
        agent.train(buf, B)
    agent.flush()
    torch.cuda.synchronize()
    rates = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(calls):
            agent.train(buf, B)
        agent.flush()
        torch.cuda.synchronize()
        rates.append(calls / (time.perf_counter() - t0))
    out = dict(arm='alone_attached_peer_preset' if attached else 'alone_single_gpu', calls=calls, train_per_s=[round(r, 1) for r in rates], median=round(float(np.median(rates)), 1),
               form=('pipe' if agent._pipe is not None else 'graph'))
    if attached:
        out['fused_groups'] = sorted(agent.core.fused_groups)
        out['status'] = int(agent.core.exchange.status(raise_on_error=False))
    del agent
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
        if arm.startswith('alone'):
            res.append(run_alone(args.workload, args.calls, args.warm, arm == 'alone_attached'))
        else:
            res.append(run_arm(args.workload, args.world, args.calls, args.warm, arm == 'attached', args.two_shot_kb))
        print(json.dumps(res[-1]), flush=True)
    by = {r['arm']: r for r in res}
    if 'alone_attached_peer_preset' in by and 'alone_single_gpu' in by:
        print(json.dumps(dict(workload=args.workload, protocol_only_attached_over_single=round(by['alone_attached_peer_preset']['median'] / by['alone_single_gpu']['median'], 4))), flush=True)
    if 'attached' in by and 'unattached' in by:
        print(json.dumps(dict(workload=args.workload, world=args.world, two_shot_kb=args.two_shot_kb,
                              attached_over_unattached=round(by['attached']['aggregate_train_per_s'] / by['unattached']['aggregate_train_per_s'], 4))), flush=True)


if __name__ == '__main__':
    main()
