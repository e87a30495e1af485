"""Several data-parallel ranks inside ONE process (rlrep_amd/comm.py LoopbackGroup: every rank's block is plain device memory, the peers are
plain pointers -- rlrep_comm_connect_local): the same device code as between processes / GPUs (csrc/dp_pull.h, csrc/comm.hip), with no IPC, no
torch.distributed and no process time-slicing.  The reference is a single process: there is nothing to compare with but arithmetic -- the sum
in RANK ORDER, bit for bit on every rank.  Each rank's launches go to a stream of their own (a rank's launch waits on the device for its
peers': they must not be queued behind it), issued by a thread of their own where the host would otherwise block behind a waiting launch."""
import os
import sys
import threading

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE, os.path.join(HERE, 'golden')]

pytestmark = pytest.mark.gpu


def _data(rank, n, call):
    rs = np.random.RandomState(1000 * call + rank)
    return (rs.standard_normal(n) * (10.0 ** rs.randint(-3, 4, size=n))).astype(np.float32)


def _want(world, n, call):
    w = _data(0, n, call).copy()
    for r in range(1, world):
        w = w + _data(r, n, call)                  # float32, rank order
    return w


def _run_ranks(world, body):
    """body(rank) on `world` threads; re-raises the first failure"""
    errs = []

    def run(r):
        try:
            body(r)
        except BaseException as e:               # noqa: BLE001
            errs.append((r, e))
    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errs:
        raise errs[0][1]


SIZES = [4, 1000, 4096 + 3, 131072, 481808]       # 481 808: the vlsac feature group at the headline dims, 16-byte multiple
OFFSETS = [0, 64, 128, 0, 16]


@pytest.mark.parametrize('world', [2, 3, 4])
def test_loopback_exchanges_are_rank_ordered_sums(world):
    """One-shot pull, two-shot (reduce-scatter + all-gather inside one launch) and the pull all-gather, `world` ranks on `world` streams of one
    GPU (HIP gives a process four concurrent hardware queues: four waiting ranks at most).  Every rank must hold exactly the rank-ordered float32 sum."""
    from rlrep_amd._lib import lib
    from rlrep_amd.comm import LoopbackGroup, _Arena, concurrent_streams
    torch.cuda.set_device(0)
    sizes = list(zip(SIZES, OFFSETS))
    grp = LoopbackGroup(world, max(SIZES) + 256, scratch_floats=world * 1024)
    for m in grp.members:
        m.set_timeout(5.0)
    streams = concurrent_streams(world)
    outs = {}
    gathered = {}

    def body(r):
        ex = grp[r]
        with torch.cuda.stream(streams[r]):
            for mode in (1, 2):
                for call, (n, off) in enumerate(sizes):
                    ex.arena[off:off + n].copy_(torch.from_numpy(_data(r, n, call + 10 * mode)).cuda(), non_blocking=True)
                    outs[(r, mode, call)] = ex.all_reduce(off, n, mode=mode, timeout_us=5_000_000)
            # all-gather of 1024-float segments in the exchange scratch (block offset = the arena's size rounded to 64 floats)
            a0 = (grp.arena_floats + 63) & ~63
            seg = torch.full((1024,), float(r + 1), device='cuda')
            sc = torch.as_tensor(_Arena(lib.rlrep_comm_scratch(ex.h), world * 1024), device='cuda:0')
            sc[r * 1024:(r + 1) * 1024].copy_(seg)
            ex.all_gather(a0, 1024)
            gathered[r] = sc.clone()
        streams[r].synchronize()

    _run_ranks(world, body)
    torch.cuda.synchronize()
    for r in range(world):
        assert grp[r].status() == 0
        for mode in (1, 2):
            for call, (n, off) in enumerate(sizes):
                assert np.array_equal(outs[(r, mode, call)].cpu().numpy(), _want(world, n, call + 10 * mode)), (world, r, mode, n)
        want = torch.cat([torch.full((1024,), float(q + 1)) for q in range(world)])
        assert torch.equal(gathered[r].cpu(), want), (world, r)
    grp.close()


def test_eight_ranks_one_after_the_other_read_fan_in_of_a_full_node():
    """world = 8: the one-shot pull's eight-way read (dp_sum4_w<8>) and its rank order.  Eight WAITING launches cannot be resident on one GPU of
    this runtime, so the ranks run one after the other on one stream, each with its peers marked as arrived (rlrep_comm_debug_preset): the data
    paths and the arithmetic of a full node, not its timing."""
    from rlrep_amd._lib import lib, check
    from rlrep_amd.comm import LoopbackGroup
    torch.cuda.set_device(0)
    world, n, off = 8, 481808, 16
    grp = LoopbackGroup(world, n + 256)
    for r in range(world):
        grp[r].arena[off:off + n].copy_(torch.from_numpy(_data(r, n, 77)).cuda())
    torch.cuda.synchronize()
    for r in range(world):
        check(lib.rlrep_comm_debug_preset(grp[r].h, 7, 1), 'debug_preset')
        got = grp[r].all_reduce(off, n, mode=1, timeout_us=2_000_000)
        torch.cuda.synchronize()
        assert grp[r].status() == 0
        assert np.array_equal(got.cpu().numpy(), _want(world, n, 77)), r
    grp.close()


def test_loopback_late_rank_times_out_skips_and_reports():
    """Rank 1 never launches: rank 0's wait runs out within the bound, the launch writes NOTHING (a timed-out launch applies nothing), the late
    rank's bit is set and status() raises."""
    from rlrep_amd.comm import LoopbackGroup
    import time
    torch.cuda.set_device(0)
    grp = LoopbackGroup(2, 4096)
    ex = grp[0]
    ex.arena[:1024].fill_(1.0)
    out = torch.full((1024,), -7.0, device='cuda')
    t0 = time.time()
    ex.all_reduce(0, 1024, out=out, timeout_us=300_000)
    torch.cuda.synchronize()
    assert time.time() - t0 < 20.0
    assert ex.status(raise_on_error=False) == 0b10
    assert torch.all(out == -7.0), 'a launch that saw a timeout must not write a partial sum'
    # ... and the rank stays poisoned until the host has seen the error: the next launch finds its peer "arrived" and still writes nothing
    from rlrep_amd._lib import lib, check
    check(lib.rlrep_comm_debug_preset(ex.h, 7, 1), 'debug_preset')
    ex.all_reduce(0, 1024, out=out, timeout_us=300_000)
    torch.cuda.synchronize()
    assert torch.all(out == -7.0), 'a rank that is out of step applies nothing more until the host has cleared the error'
    with pytest.raises(RuntimeError, match='did not arrive'):
        ex.status(clear=True)
    assert ex.status() == 0
    check(lib.rlrep_comm_debug_preset(ex.h, 7, 1), 'debug_preset')
    ex.all_reduce(0, 1024, out=out, timeout_us=300_000)
    torch.cuda.synchronize()
    assert ex.status() == 0 and torch.all(out == 1.0), 'cleared: the exchange works again (the absent peer\'s arena holds zeros)'
    grp.close()


def _make_loopback_agents(case_name, world, enable=None, graph=False):
    from fixture_io import Case
    from test_hip_parity import make_agent, make_buffer
    from rlrep_amd.comm import LoopbackGroup
    c = Case(case_name)
    old = os.environ.get('RLREP_ENABLE')
    if enable is not None:
        os.environ['RLREP_ENABLE'] = enable
    try:
        grp = LoopbackGroup(world)              # (sized by the first agent that joins)
        grp.timeout_s = 5.0
        agents = [make_agent(c, seed=17, loopback=(grp, r)) if graph else make_agent(c, loopback=(grp, r)) for r in range(world)]
        bufs = [make_buffer(c) for _ in range(world)]
    finally:
        if enable is not None:
            if old is None:
                os.environ.pop('RLREP_ENABLE', None)
            else:
                os.environ['RLREP_ENABLE'] = old
    return c, grp, agents, bufs


@pytest.mark.parametrize('case,world', [('sac_tiny', 3), ('vlsac_tiny', 3), ('spedersac_tiny', 3), ('ctrlsac_tiny', 3), ('spedersac_tiny', 4), ('ctrlsac_tiny', 4)])
def test_loopback_ranks_two_shot_equals_one_shot_and_the_oracle(case, world):
    """Three / four replicas in one process (four = BASELINE config 4's rank count), eager step programs, injected draws: two-shot forced for every
    slice == one-shot, bit for bit (both are the rank-ordered sum), replicas bit-identical, and equal (1e-4) to the CPU oracle on the concatenated
    global batch -- spedersac's Phibar / v and ctrlsac's in-batch negatives over ALL ranks' minibatches included."""
    from fixture_io import Case, rel_l2
    from test_dp import _inputs, _oracle_global
    trains, states = 2, {}
    for form, enable in (('two_shot', 'dp_two_shot_kb=0.001'), ('one_shot', 'dp_two_shot_kb=0')):
        c, grp, agents, bufs = _make_loopback_agents(case, world, enable)
        assert all(a.core.fused_groups for a in agents) and all(a.core.feature_exchange_count() == 0 for a in agents)
        rs = np.random.RandomState(11)
        per_call = [_inputs(c, rs, world) for _ in range(trains)]
        from rlrep_amd.comm import concurrent_streams
        streams = concurrent_streams(world)

        def body(r):
            with torch.cuda.stream(streams[r]):
                for t in range(trains):
                    agents[r].train_injected(bufs[r], c.B, *per_call[t][r])
            streams[r].synchronize()
        _run_ranks(world, body)
        torch.cuda.synchronize()
        for a in agents:
            assert a.core.exchange.status() == 0
        st = [{k: v.numpy() for k, v in a.core.state().items()} for a in agents]
        for r in range(1, world):
            for k, v in st[0].items():
                assert np.array_equal(v, st[r][k]), f'replicas diverged at {k} ({form}, rank {r})'
        states[form] = st[0]
        del agents
        grp.close()
    for k, v in states['two_shot'].items():
        assert np.array_equal(v, states['one_shot'][k]), f'two-shot != one-shot at {k}'
    c = Case(case)
    rs = np.random.RandomState(11)
    per_call = [_inputs(c, rs, world) for _ in range(trains)]
    P = _oracle_global(c, per_call, trains=trains).state()
    for k, v in states['two_shot'].items():
        if k in P and not k.endswith('noise'):
            assert rel_l2(v, P[k].numpy()) < 1e-4, k


def test_attach_leaves_out_groups_that_could_starve_the_other_chain():
    """The co-residency bound of the two-chain train() (csrc/dp_pull.h, "Progress with SEVERAL channels in flight"): the waiting blocks of the two
    largest attached optimizer launches must fit the chip together.  With the size cap lifted (RLREP_ENABLE=dp_fused_mb=64) ctrlsac at main.py's
    dimensions has a 25 MB feature group -- 6 400 blocks of the 1 280 resident ones: it must stay with the caller's all-reduce, the critic and the
    actor group are attached; at the default cap nothing changes for the headline agent (all three groups attached)."""
    import bench
    from rlrep_amd.comm import LoopbackGroup
    old = os.environ.get('RLREP_ENABLE')
    try:
        os.environ['RLREP_ENABLE'] = 'dp_fused_mb=64'
        alg, S, A, B, kw = bench.WORKLOADS['ctrlsac_halfcheetah_f2048_b256']
        grp = LoopbackGroup(2)
        agents = [bench.make_agent(alg, S, A, B, {**kw, 'loopback': (grp, r)}) for r in range(2)]
        lay = agents[0].core.layout
        assert lay.group_floats[0] * 4 > 20e6
        assert 0 not in agents[0].core.fused_groups and {2} <= set(agents[0].core.fused_groups), agents[0].core.fused_groups
        assert agents[0].core.fused_groups == agents[1].core.fused_groups
        assert not agents[0]._fused_all
        del agents
        grp.close()
    finally:
        if old is None:
            os.environ.pop('RLREP_ENABLE', None)
        else:
            os.environ['RLREP_ENABLE'] = old
    alg, S, A, B, kw = bench.WORKLOADS['vlsac_halfcheetah_f256_b256']
    grp = LoopbackGroup(2)
    agents = [bench.make_agent(alg, S, A, B, {**kw, 'loopback': (grp, r)}) for r in range(2)]
    assert set(agents[0].core.fused_groups) == {0, 1, 2} and agents[0]._fused_all
    del agents
    grp.close()
