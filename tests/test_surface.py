"""CPU tests of the drop-in surface against tables captured from the reference by tests/golden/make_surface.py:
constructor / method signatures through `rlrep_amd.dropin`'s module aliases, state_dict keys + forward known-answers of the
signature-only modules (rows n1-n5, a2, a4-a6, b1-b3), the diffsrsac noise schedule (e2) and the ReplayBuffer ring semantics (f1)."""
import inspect
import json
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SURF = os.path.join(HERE, 'golden', 'surface')
SIGS = json.load(open(os.path.join(SURF, 'signatures.json')))
Z = np.load(os.path.join(SURF, 'modules.npz'), allow_pickle=False)
META = json.loads(str(Z['meta/json']))


def _resolve(path):
    import importlib
    import rlrep_amd.dropin  # noqa: F401  (aliases utils / agent / networks to rlrep_amd's packages)
    mod, name = path.rsplit('.', 1)
    return getattr(importlib.import_module(mod), name)


def _check_sig(path, ref_params, fn):
    mine = list(inspect.signature(fn).parameters.items())
    names = [n for n, _ in mine]
    pos = 0
    for n, kind, default in ref_params:
        if 'VAR_' in kind:
            continue
        assert n in names, f'{path}: parameter {n!r} of the reference is missing ({names})'
        p = dict(mine)[n]
        # same position among the positional parameters, same default
        assert names.index(n) == pos, f'{path}: parameter {n!r} at position {names.index(n)}, reference has it at {pos}'
        pos += 1
        d = None if p.default is inspect._empty else repr(p.default)
        if default is None:
            assert d is None, f'{path}: {n!r} is required in the reference, has default {d} here'
        else:
            assert d is not None and eval(d) == eval(default), f'{path}: default of {n!r} is {d}, reference {default}'
    # anything this build adds must be optional
    ref_names = {n for n, _, _ in ref_params}
    for n, p in mine:
        if n not in ref_names:
            assert p.default is not inspect._empty or p.kind in (p.VAR_KEYWORD, p.VAR_POSITIONAL), f'{path}: extra required parameter {n!r}'


@pytest.mark.parametrize('path', sorted(SIGS))
def test_signatures_match_the_reference(path):
    entry = SIGS[path]
    obj = _resolve(path)
    if isinstance(entry, list):                      # a plain function
        _check_sig(path, entry, obj)
        return
    if '_fields' in entry:
        assert list(obj._fields) == entry['_fields']
        return
    for meth, ref in entry.items():
        assert hasattr(obj, meth), f'{path}.{meth} is missing'
        if ref == 'property':
            assert isinstance(inspect.getattr_static(obj, meth), property), f'{path}.{meth} must be a property'
            continue
        _check_sig(f'{path}.{meth}', ref, getattr(obj, meth))


def _sd(tag):
    pre = f'{tag}/sd/'
    return {k[len(pre):]: torch.from_numpy(Z[k]) for k in Z.files if k.startswith(pre)}


class _Space:
    low = np.array([-1.0, -2.0, -2.0], np.float32)
    high = np.array([2.0, 2.0, 2.0], np.float32)


MODULE_TAGS = [t for t in META if '.' in t]


@pytest.mark.parametrize('tag', MODULE_TAGS)
def test_module_state_dict_and_forward_match_the_reference(tag):
    m = META[tag]
    cls = _resolve(tag)
    kw = dict(m['ctor'])
    if tag.endswith('GaussianPolicy'):
        kw['action_space'] = _Space()
    mod = cls(**kw)
    assert list(mod.state_dict().keys()) == m['sd_keys'], (list(mod.state_dict().keys()), m['sd_keys'])
    mod.load_state_dict(_sd(tag), strict=True)
    ins = [torch.from_numpy(Z[f'{tag}/in/{i}']) for i in range(m['n_in'])]
    with torch.no_grad():
        res = getattr(mod, m['call'])(*ins)
    res = res if isinstance(res, (tuple, list)) else (res,)
    assert len(res) == m['n_out']
    for i, r in enumerate(res):
        assert np.allclose(r.numpy(), Z[f'{tag}/out/{i}'], rtol=1e-5, atol=1e-6), (tag, i)


def test_rff_linear_critic_initialisation_quirk():
    """networks/critic.py:129-136: l1 and l4 start from the SAME W ~ N(0,1), b ~ U(0, 2*3.1415926)."""
    cls = _resolve('networks.critic.RFFLinearCritic')
    torch.manual_seed(0)
    c = cls(9, 20, 12)
    assert torch.equal(c.l1.weight, c.l4.weight) and torch.equal(c.l1.bias, c.l4.bias)
    assert META['networks.critic.RFFLinearCritic']['l1_equals_l4']
    assert 0.0 <= float(c.l1.bias.min()) and float(c.l1.bias.max()) <= 2 * 3.1415926
    assert abs(float(c.l1.weight.std()) - 1.0) < 0.2


def test_gaussian_policy_sample_with_pinned_noise():
    import torch.distributions.normal as tdn
    cls = _resolve('networks.policy.GaussianPolicy')
    pol = cls(action_space=_Space(), **META['policy_sample']['ctor'])
    pol.load_state_dict(_sd('policy_sample'), strict=True)
    eps = torch.from_numpy(Z['policy_sample/eps'])
    orig_sn, orig_rl = tdn._standard_normal, torch.randn_like
    tdn._standard_normal = lambda shape, dtype, device: eps.clone()
    torch.randn_like = lambda t, **k: eps.clone()
    try:
        with torch.no_grad():
            out = pol.sample(torch.from_numpy(Z['policy_sample/in/0']))
    finally:
        tdn._standard_normal, torch.randn_like = orig_sn, orig_rl
    for i, r in enumerate(out):
        assert np.allclose(r.numpy(), Z[f'policy_sample/out/{i}'], rtol=1e-5, atol=1e-5), i


def test_live_actor_distribution_with_pinned_noise():
    """agent/sac/actor.py:16-91: DiagGaussianActor -> SquashedNormal: mean, rsample, log_prob."""
    import torch.distributions.normal as tdn
    cls = _resolve('agent.sac.actor.DiagGaussianActor')
    act = cls(**META['actor']['ctor'])
    assert list(act.state_dict().keys()) == META['actor']['sd_keys']
    act.load_state_dict(_sd('actor'), strict=True)
    eps = torch.from_numpy(Z['actor/eps'])
    orig = tdn._standard_normal
    tdn._standard_normal = lambda shape, dtype, device: eps.clone()
    try:
        with torch.no_grad():
            d = act(torch.from_numpy(Z['actor/in/0']))
            y = d.rsample()
            lp = d.log_prob(y).sum(-1, keepdim=True)
            mu = d.mean
    finally:
        tdn._standard_normal = orig
    for got, i in ((mu, 0), (y, 1), (lp, 2)):
        assert np.allclose(got.numpy(), Z[f'actor/out/{i}'], rtol=1e-5, atol=1e-5), i


def test_alphabars_both_paths_match_the_reference():
    """diffsrsac_agent.py:178-203 (e2): the scipy path and the scipy-free quadrature against the reference's own table, and against
    the table the diffsrsac_hc fixture carries."""
    from rlrep_amd.agent.diffsrsac import diffsrsac_agent as d
    want = Z['alphabars/default']
    got = d.generate_alphabars(0.3, 0.1, 1000)
    assert got.dtype == np.float32 and np.array_equal(got, want)
    x = np.linspace(0, 1, 1000)
    raw = 1.0 - d._beta_cdf(x, 0.3, 0.1)
    alt = np.clip(raw, a_min=raw[-2], a_max=raw[1]).astype(np.float32)
    assert np.max(np.abs(alt - want)) < 2e-6, np.max(np.abs(alt - want))
    hc = np.load(os.path.join(HERE, 'golden', 'diffsrsac_hc.npz'))['init/noise_alphabars']
    assert np.array_equal(hc.reshape(-1), want.reshape(-1))
    ab, al = d.DIFFSRSACAgent.generate_alphabars_and_alphas(0.3, 0.1, 1000)
    assert np.array_equal(np.asarray(ab, np.float32).reshape(-1), want.reshape(-1))
    if 'alphas/default' in Z.files:
        assert np.allclose(np.asarray(al, np.float64).reshape(-1), Z['alphas/default'].reshape(-1), rtol=1e-5, atol=1e-7)


def test_replay_ring_wraps_like_the_reference():
    """utils/buffer.py:28-36: 8 adds into a ring of 5 -> ptr 3, size 5, rows 5,6,7 overwrite slots 0,1,2 (pinned staging, two-piece
    flush across the wrap).  Runs on the CPU device path of the same class."""
    from rlrep_amd.utils.buffer import ReplayBuffer
    m = META['ring']
    for stage_rows in (4096, 3, 1):                 # staging buffer larger than / smaller than the ring: flushes mid-way and across the wrap
        rb = ReplayBuffer(2, 1, max_size=m['max_size'], device='cpu', stage_rows=stage_rows)
        for i in range(m['adds']):
            rb.add(np.full(2, i, np.float32), np.full(1, 10 + i, np.float32), np.full(2, 100 + i, np.float32), float(i), float(i % 2))
        assert (rb.ptr, rb.size, rb.max_size) == (m['ptr'], m['size'], m['max_size'])
        for k in ('state', 'action', 'next_state', 'reward', 'done'):
            assert np.array_equal(getattr(rb, k), Z[f'ring/{k}']), (stage_rows, k)
        b = rb.sample(4)
        assert tuple(b._fields) == ('state', 'action', 'reward', 'next_state', 'done')
        assert b.state.shape == (4, 2) and b.reward.shape == (4, 1) and b.state.dtype == torch.float32


def test_replay_ring_round_robin_sharding():
    """the replay partitioned by transition index -- W shards offered the same stream hold, together, exactly the single
    ring's content (transition i at slot i // W of shard i % W), also across the wrap-around."""
    from rlrep_amd.utils.buffer import ReplayBuffer
    W, cap, n = 3, 4, 17                      # 17 transitions into 3 shards of 4 slots (global capacity 12): the oldest 5 are overwritten
    shards = [ReplayBuffer(2, 1, max_size=cap, device='cpu', shard=(r, W)) for r in range(W)]
    for i in range(n):
        for s in shards:
            s.add(np.full(2, i, np.float32), np.full(1, i, np.float32), np.full(2, -i, np.float32), float(i), 0.0)
    kept = set()
    for r, s in enumerate(shards):
        assert s.size == cap
        vals = s.state[:, 0].astype(int).tolist()
        for slot, v in enumerate(vals):
            assert v % W == r and (v // W) % cap == slot, (r, slot, v)
        kept.update(vals)
    assert kept == set(range(n - W * cap, n))


def test_info_dict_value_types_match_the_reference():
    """sac_agent.py:154-166 (quirk Q14): `alpha_loss` and `alpha` are 0-dim tensors (fp32 / fp64), every other metric a Python float."""
    from rlrep_amd.core import LazyInfo
    names = ['q_loss', 'alpha_loss', 'alpha', '']
    info = LazyInfo(names, torch.tensor([1.5, 0.25, 0.1, 9.0]))
    assert isinstance(info['q_loss'], float)
    assert torch.is_tensor(info['alpha_loss']) and info['alpha_loss'].dtype == torch.float32 and info['alpha_loss'].ndim == 0
    assert torch.is_tensor(info['alpha']) and info['alpha'].dtype == torch.float64
    assert abs(float(info['alpha']) - 0.1) < 1e-7 and set(info.keys()) == {'q_loss', 'alpha_loss', 'alpha'}
