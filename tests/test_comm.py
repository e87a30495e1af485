"""One-shot all-reduce over hipIpc-mapped inboxes (rlrep_amd/csrc/comm.hip; SURVEY.md 5.8 / 8e: the latency-shaped gradient all-reduce, K17).
The reference is a single process: there is nothing to compare with but arithmetic -- the sum in RANK ORDER, which every rank must produce
bit for bit.  Several processes share the one GPU of the test box and map each other's inbox (the multi-GPU node only changes the wires)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _data(rank, n, call):
    rs = np.random.RandomState(1000 * call + rank)
    return (rs.standard_normal(n) * (10.0 ** rs.randint(-3, 4, size=n))).astype(np.float32)


SIZES = [1, 3, 64, 1000, 4096 + 3, 262144, 481810]        # ... 481 810: the vlsac feature group at the headline dims (1.93 MB)


def _worker(rank, world, port, q):
    try:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
        torch.cuda.set_device(0)
        from rlrep_amd.comm import OneShotAllReduce
        comm = OneShotAllReduce(max(SIZES))
        out = []
        for call, n in enumerate(SIZES * 2):                   # twice: both epoch parities of every size, back to back without a host sync
            t = torch.from_numpy(_data(rank, n, call)).cuda()
            comm.all_reduce(t)
            out.append(t)
        comm.check()
        res = [t.cpu().numpy() for t in out]
        # a rank that runs ahead: rank 0 starts the next all-reduce immediately, the others late -- the bounded wait must simply wait
        if rank != 0:
            torch.cuda.synchronize(); import time; time.sleep(0.3)
        t = torch.full((1024,), float(rank + 1), device='cuda')
        comm.all_reduce(t)
        comm.check()
        res.append(t.cpu().numpy())
        q.put((rank, res, comm.fine_grained))
        dist.barrier()
        comm.close()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc(), None))


@pytest.mark.parametrize('world', [2, 3])
def test_one_shot_allreduce_is_the_rank_ordered_sum_on_every_rank(world):
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r = q.get(timeout=300)
        assert isinstance(r[1], list), r[1]
        res[r[0]] = r[1]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for call, n in enumerate(SIZES * 2):
        want = _data(0, n, call).copy()
        for r in range(1, world):
            want = want + _data(r, n, call)                     # float32, rank order: ((r0 + r1) + r2)
        for r in range(world):
            assert np.array_equal(res[r][call], want), (world, call, n, r)
    for r in range(world):
        assert np.array_equal(res[r][-1], np.full(1024, world * (world + 1) / 2, np.float32))


def _agent_worker(rank, world, port, q, oneshot):
    try:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if oneshot:
            os.environ['RLREP_ONESHOT_ALLREDUCE'] = '1'
        dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
        torch.cuda.set_device(0)
        from fixture_io import Case
        from test_hip_parity import make_agent, make_buffer
        from test_dp import _inputs
        c = Case('vlsac_tiny')
        agent = make_agent(c)
        buf = make_buffer(c)
        rs = np.random.RandomState(11)
        for t in range(2):
            per_rank = _inputs(c, rs, world)
            agent.train_injected(buf, c.B, *per_rank[rank])
        torch.cuda.synchronize()
        if oneshot:
            assert agent._oneshot_comm is not None
            agent._oneshot_comm.check()
        q.put((rank, {k: v.numpy() for k, v in agent.core.state().items()}, None))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc(), None))


def test_data_parallel_train_over_the_one_shot_allreduce_equals_gloo():
    """RLREP_ONESHOT_ALLREDUCE=1: two ranks' vlsac train() with every gradient slice summed by the one-shot all-reduce end in exactly the state of
    the same run over gloo (two ranks: a + b in either order), replicas bit-identical."""
    out = {}
    for oneshot in (True, False):
        world, port = 2, _free_port()
        ctx = mp.get_context('spawn')
        q = ctx.Queue()
        procs = [ctx.Process(target=_agent_worker, args=(r, world, port, q, oneshot)) for r in range(world)]
        for p in procs:
            p.start()
        res = {}
        for _ in range(world):
            r = q.get(timeout=400)
            assert isinstance(r[1], dict), r[1]
            res[r[0]] = r[1]
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
        for k, v in res[0].items():
            assert np.array_equal(v, res[1][k]), f'replicas diverged at {k} (oneshot={oneshot})'
        out[oneshot] = res[0]
    for k, v in out[True].items():
        assert np.array_equal(v, out[False][k]), f'one-shot != gloo at {k}'
