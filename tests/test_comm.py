"""The data-parallel exchanges INSIDE the launches (rlrep_amd/csrc/comm.hip, dp_pull.h, rlrep_amd/comm.py): the gradient
sums in the optimizer launches (one-shot pull, two-shot reduce-scatter + all-gather), spedersac's pushed Phibar / v, ctrlsac's pull gather /
reduce-scatter.  The reference is a single process: there is nothing to compare with but arithmetic -- the sum in RANK ORDER, which every rank must
produce bit for bit -- and the same train() over torch.distributed (gloo) summed in rank order, which must end in exactly the same state.
Several processes share the one GPU of the test box and map each other's block over hipIpc (a multi-GPU node only changes the wires: there the
block is fine-grained memory); tests/test_loopback.py runs the same device code with all ranks in one process."""
import os
import socket
import sys
import time

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE, os.path.join(HERE, 'golden')]

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _data(rank, n, call):
    rs = np.random.RandomState(1000 * call + rank)
    return (rs.standard_normal(n) * (10.0 ** rs.randint(-3, 4, size=n))).astype(np.float32)


SIZES = [1, 3, 64, 1000, 4096 + 3, 262144, 481810]        # ... 481 810: the vlsac feature group at the headline dims (1.93 MB)
OFFSETS = [0, 5, 64, 0, 128, 0, 16]                        # (odd offsets take the element-wise path)


def _spawn(fn, world, *args, timeout=400):
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_entry, args=(r, world, port, q, fn) + args) for r in range(world)]
    for p in procs:
        p.daemon = True
        p.start()
    res = {}
    try:
        for _ in range(world):
            r = q.get(timeout=timeout)
            assert not isinstance(r[1], str), r[1]
            res[r[0]] = r[1:]
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    finally:
        for p in procs:                 # (a rank that failed leaves its peers in a collective: never let them outlive the test)
            if p.is_alive():
                p.terminate()
                p.join(timeout=10)
                if p.is_alive():
                    p.kill()
    return res


def _entry(rank, world, port, q, body, *args):
    """Process entry point (top level: the spawn context pickles it by name); `body` names one of the *_body functions below."""
    try:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
        torch.cuda.set_device(0)
        globals()[body](rank, world, q, *args)
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc()))


def _pull_body(rank, world, q):
    from rlrep_amd.comm import GradientExchange
    ex = GradientExchange(max(SIZES) + 256)
    assert ex.usable and ex.same_device
    assert ex.probe()
    out = []
    for call, (n, off) in enumerate(list(zip(SIZES, OFFSETS)) * 2):           # twice over the same addresses, back to back without a host sync
        ex.arena[off:off + n].copy_(torch.from_numpy(_data(rank, n, call)))
        # first pass: one-shot pull; second pass: two-shot (reduce-scatter + all-gather in one launch; world >= 3 and 16-byte ranges, else the
        # library falls back to the one-shot form by itself)
        out.append(ex.all_reduce(off, n, mode=1 if call < len(SIZES) else 2))
    torch.cuda.synchronize()
    assert ex.status() == 0
    res = [t.cpu().numpy() for t in out]
    # a rank that runs ahead: rank 0 starts the next exchange immediately, the others late -- the bounded wait must simply wait
    if rank != 0:
        time.sleep(0.3)
    ex.arena[:1024].fill_(float(rank + 1))
    res.append(ex.all_reduce(0, 1024).cpu().numpy())
    assert ex.status() == 0
    q.put((rank, res, ex.fine_grained))
    dist.barrier()
    ex.close()



@pytest.mark.parametrize('world', [2, 3, 4])
def test_in_launch_exchange_is_the_rank_ordered_sum_on_every_rank(world):
    res = _spawn('_pull_body', world)
    for call, (n, off) in enumerate(list(zip(SIZES, OFFSETS)) * 2):
        want = _data(0, n, call).copy()
        for r in range(1, world):
            want = want + _data(r, n, call)                     # float32, rank order: ((r0 + r1) + r2)
        for r in range(world):
            assert np.array_equal(res[r][0][call], want), (world, call, n, r)
    for r in range(world):
        assert np.array_equal(res[r][0][-1], np.full(1024, world * (world + 1) / 2, np.float32))
        assert res[r][1], 'the shared block should be fine-grained device memory on this runtime (hipExtMallocWithFlags + hipIpcGetMemHandle)'


def _late_body(rank, world, q):
    """Rank 1 never calls: rank 0's bounded wait runs out, sets rank 1's bit, the launch drains and status() raises -- no hang, no re-exec."""
    from rlrep_amd.comm import GradientExchange
    ex = GradientExchange(4096)
    mask, raised = 0, False
    if rank == 0:
        ex.arena[:1024].fill_(1.0)
        t0 = time.time()
        ex.all_reduce(0, 1024, timeout_us=300000)
        torch.cuda.synchronize()
        took = time.time() - t0
        mask = ex.status(raise_on_error=False)
        try:
            ex.status(clear=True)
        except RuntimeError as e:
            raised = 'did not arrive' in str(e)
        assert ex.status() == 0                                  # cleared
        q.put((rank, mask, raised, took))
    else:
        q.put((rank, 0, True, 0.0))
    dist.barrier()
    ex.close()



def test_a_rank_that_never_arrives_sets_the_mask_and_raises_within_the_timeout():
    res = _spawn('_late_body', 2)
    mask, raised, took = res[0]
    assert mask == 0b10 and raised, (mask, raised)
    assert took < 30.0, took


def _ordered_all_reduce():
    """torch.distributed.all_reduce(SUM) with the sum formed in RANK order on every rank (gloo's ring order is not the rank order): what the
    in-launch exchange computes, so that the two forms can be compared bit for bit with any number of ranks."""
    real = dist.all_reduce

    def ordered(t, *a, op=dist.ReduceOp.SUM, group=None, async_op=False, **k):
        if op != dist.ReduceOp.SUM or not t.is_floating_point() or async_op:
            return real(t, *a, op=op, group=group, async_op=async_op, **k)
        parts = [torch.empty_like(t) for _ in range(dist.get_world_size(group))]
        dist.all_gather(parts, t.contiguous(), group=group)
        acc = parts[0].clone()
        for p in parts[1:]:
            acc = acc + p
        t.copy_(acc)
    dist.all_reduce = ordered


def _agent_body(rank, world, q, case, fused, graph, calls, pipe_dp=False, enable='', ordered=False, pipeline=True):
    os.environ['RLREP_DP_FUSED'] = '1' if fused else '0'
    os.environ['RLREP_PIPELINE_DP'] = '1' if pipe_dp else '0'
    os.environ['RLREP_ENABLE'] = ','.join(t for t in (enable, 'dp_timeout_s=20') if t)      # (a protocol bug must fail the test in seconds, not in the watchdog's minutes)
    if not pipeline:
        os.environ['RLREP_PIPELINE'] = '0'
    if ordered:
        _ordered_all_reduce()
    from fixture_io import Case
    from test_hip_parity import make_agent, make_buffer
    from test_dp import _inputs
    c = Case(case)
    agent = make_agent(c, seed=17) if graph else make_agent(c)
    assert bool(agent.core.fused_groups) == fused, agent.core.fused_groups
    buf = make_buffer(c)
    took = None
    if graph:
        agent.use_graph = True
        for t in range(calls):
            info = agent.train(buf, c.B)
            if t == calls // 2:
                agent.select_action(np.zeros(c.S, np.float32))
            if rank == 0 and t == 1:
                float(next(iter(info.values())))             # ONE rank reads its metrics (a flush): no collective behind it, the ranks stay in step
        agent.flush()
        took = ('pipe' if agent._pipe.get('mode') != 3 else 'pipe_dp') if agent._pipe is not None else ('graph' if not isinstance(agent._graph, list) else 'segments:%d' % sum(1 for k, _ in agent._graph if k == 'coll'))
    else:
        rs = np.random.RandomState(11)
        for t in range(calls):
            per_rank = _inputs(c, rs, world)
            agent.train_injected(buf, c.B, *per_rank[rank])
    torch.cuda.synchronize()
    if fused:
        assert agent.core.exchange.status() == 0
    st = {k: v.numpy() for k, v in agent.core.state().items()}
    q.put((rank, st, took, sorted(agent.core.fused_groups), agent.core.feature_exchange_count()))
    dist.barrier()
    del agent



@pytest.mark.parametrize('case,world', [('vlsac_tiny', 2), ('spedersac_tiny', 2), ('ctrlsac_tiny', 2), ('sac_tiny', 3), ('spedersac_tiny', 3), ('ctrlsac_tiny', 3)])
def test_train_with_exchanges_inside_the_launches_equals_gloo(case, world):
    """Eager step programs, injected draws: every gradient slice summed inside its optimizer launch AND the batch-coupled exchanges of the
    feature step inside the step program (spedersac: Phibar / v pushed by the column-sum launches and summed by their consumers; ctrlsac: one pull
    launch each for the all-gather of mu(s') and the reduce-scatter of its gradient) ends in EXACTLY the state of the same run with
    torch.distributed collectives between the launches, summed in rank order -- bit for bit with two and with three ranks; replicas bit-identical."""
    out = {}
    for fused in (True, False):
        res = _spawn('_agent_body', world, case, fused, False, 2, False, '', True)
        for r in range(1, world):
            for k, v in res[0][0].items():
                assert np.array_equal(v, res[r][0][k]), f'replicas diverged at {k} (fused={fused}, rank {r})'
        out[fused] = res[0]
    assert out[True][2], 'no group was attached'
    assert out[True][3] == 0, 'a batch-coupled exchange was left outside the launches'
    if case.startswith(('spedersac', 'ctrlsac')):
        assert out[False][3] > 0
    for k, v in out[True][0].items():
        assert np.array_equal(v, out[False][0][k]), f'in-launch exchanges != rank-ordered gloo at {k}'


@pytest.mark.parametrize('case', ['sac_tiny', 'vlsac_tiny'])
def test_two_shot_in_the_optimizer_launch_is_the_rank_ordered_sum_bit_for_bit(case):
    """Three ranks: the two-shot form inside the optimizer launches (forced for every slice: RLREP_ENABLE=dp_two_shot_kb=0.001 -- reduce-scatter
    into the reduced region, RED handshake, all-gather from the shards' owners) == the one-shot pull == gloo summed in rank order."""
    world, out = 3, {}
    for form, (fused, enable) in (('two_shot', (True, 'dp_two_shot_kb=0.001')), ('one_shot', (True, 'dp_two_shot_kb=0')), ('gloo', (False, ''))):
        res = _spawn('_agent_body', world, case, fused, False, 3, False, enable, True)
        for r in range(1, world):
            for k, v in res[0][0].items():
                assert np.array_equal(v, res[r][0][k]), f'replicas diverged at {k} ({form}, rank {r})'
        out[form] = res[0][0]
    for form in ('one_shot', 'gloo'):
        for k, v in out['two_shot'].items():
            assert np.array_equal(v, out[form][k]), f'two-shot != {form} at {k}'


@pytest.mark.parametrize('case,world,pipeline', [('spedersac_ant512', 2, True), ('ctrlsac_hc256', 2, True), ('vlsac_hc', 3, False)])
def test_default_graph_forms_with_exchanges_inside_equal_rank_ordered_gloo_segments(case, world, pipeline):
    """BASELINE's configs 4 / 3 / 2 at their dimensions, data parallel, DEFAULT train(): with every exchange inside the launches the agents take
    the single-GPU graph forms (two chains on two streams: spedersac and ctrlsac keep their deferred step programs now) and end in exactly the
    state of graph segments around gloo collectives summed in rank order.  vlsac_hc with three ranks: its 1.93 MB feature slice and 1.05 MB
    critic slice take the two-shot form inside the optimizer launch -- as ONE graph per train() (RLREP_PIPELINE=0): three ranks share the test
    box's one GPU, and the waiting blocks of three ranks' two chains (3 x (472 + 265) of 1 280 resident) would leave no room for the launches
    they wait for; between GPUs every rank has a chip of its own (csrc/dp_pull.h, "Progress with SEVERAL channels in flight")."""
    out = {}
    for form, fused in (('fused', True), ('segments', False)):
        res = _spawn('_agent_body', world, case, fused, True, 5, False, '', True, pipeline, timeout=300)
        for r in range(1, world):
            for k, v in res[0][0].items():
                assert np.array_equal(v, res[r][0][k]), f'replicas diverged at {k} ({form}, rank {r})'
        out[form] = res[0]
    assert out['fused'][1] == ('pipe' if pipeline else 'graph'), out['fused'][1]
    assert out['fused'][3] == 0 and out['segments'][1].startswith('segments:'), (out['fused'][3], out['segments'][1])
    for k, v in out['fused'][0].items():
        assert np.array_equal(v, out['segments'][0][k]), f'exchanges inside the launches != rank-ordered gloo segments at {k}'


@pytest.mark.parametrize('case', ['vlsac_tiny', 'vlsac_hc'])
def test_default_graph_forms_with_two_ranks_equal_the_gloo_forms(case):
    """The default train() with two ranks: with the exchange inside the optimizer launches it is the single-GPU form -- two chains on two streams,
    whole hipGraphs, no collective anywhere -- and ends in exactly the state of BOTH torch.distributed forms: graph segments around six gloo
    all-reduces (sequential), and the two-chain form with one process group per chain (RLREP_PIPELINE_DP=1); select_action and a one-sided
    metrics read in between.  vlsac_hc = the headline dimensions."""
    out = {}
    for form, (fused, pipe_dp) in (('fused', (True, False)), ('segments', (False, False)), ('pipe_dp', (False, True))):
        res = _spawn('_agent_body', 2, case, fused, True, 6, pipe_dp, timeout=600)
        for k, v in res[0][0].items():
            assert np.array_equal(v, res[1][0][k]), f'replicas diverged at {k} ({form})'
        out[form] = res[0]
    assert out['fused'][1] == 'pipe', out['fused'][1]
    assert out['segments'][1].startswith('segments:6'), out['segments'][1]
    assert out['pipe_dp'][1] == 'pipe_dp', out['pipe_dp'][1]
    for form in ('segments', 'pipe_dp'):
        for k, v in out['fused'][0].items():
            assert np.array_equal(v, out[form][0][k]), f'in-launch exchange != gloo {form} at {k}'


def test_four_ranks_match_the_global_batch_oracle():
    """Four processes on the one GPU, every gradient slice summed inside its optimizer launch: replicas bit-identical, and equal (1e-4) to the
    CPU oracle run on the concatenated global batch of 4 B rows -- the rank-ordered sum of four arenas against autograd on one batch."""
    from fixture_io import Case, rel_l2
    from test_dp import _inputs, _oracle_global
    world, case = 4, 'vlsac_tiny'
    res = _spawn('_agent_body', world, case, True, False, 2, timeout=600)
    for r in range(1, world):
        for k, v in res[0][0].items():
            assert np.array_equal(v, res[r][0][k]), f'replicas diverged at {k} (rank {r})'
    c = Case(case)
    rs = np.random.RandomState(11)
    per_rank = [_inputs(c, rs, world) for _ in range(2)]
    P = _oracle_global(c, per_rank, trains=2).state()
    for k, v in res[0][0].items():
        if k in P and not k.endswith('noise'):
            assert rel_l2(v, P[k].numpy()) < 1e-4, k
