"""CPU-only: the C-ABI library loads and exports every symbol include/rlrep.h declares; layout queries
(host-only code paths, no GPU call) agree with the oracle's shape table."""
import ctypes as C
import pytest


def test_exports_every_declared_symbol():
    from rlrep_amd import _lib
    names = _lib.declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(_lib.lib, n), n
        assert n in _lib.SIGNATURES, f'{n} declared in rlrep.h but not bound in _lib.py'
    assert _lib.lib.rlrep_abi_version() == 4


@pytest.mark.parametrize('alg,kw', [
    ('sac', dict(hidden_dim=256)),
    ('vlsac', dict(hidden_dim=256, feature_dim=256, vae_hidden=256)),
    ('ctrlsac', dict(hidden_dim=64, feature_dim=128, phi_hidden_dim=64, mu_hidden_dim=64, phi_hidden_depth=2, mu_hidden_depth=2)),
    ('spedersac', dict(hidden_dim=256, critic_and_actor_hidden_dim=256, feature_dim=128, phi_hidden_dim=64, mu_hidden_dim=64,
                       phi_hidden_depth=1, mu_hidden_depth=0)),
    ('diffsrsac', dict(hidden_dim=256, feature_dim=32, phi_hidden_dim=64, nabla_mu_hidden_dim=48, phi_hidden_depth=1,
                       nabla_mu_hidden_depth=1, num_noise=1000)),
])
def test_layout_matches_shape_table(alg, kw):
    from rlrep_amd import _lib
    from oracle.shapes import param_shapes
    S, A = 17, 6
    d = _lib.Dims()
    d.alg = _lib.ALG[alg]
    d.state_dim, d.action_dim, d.hidden_dim, d.actor_hidden_dim = S, A, kw.get('hidden_dim', 256), 256
    d.feature_dim, d.vae_hidden_dim, d.num_noise, d.max_batch = kw.get('feature_dim', 0), kw.get('vae_hidden', 0), kw.get('num_noise', 20), 256
    d.phi_hidden_dim, d.phi_hidden_depth = kw.get('phi_hidden_dim', 0), kw.get('phi_hidden_depth', 0)
    d.mu_hidden_dim = kw.get('mu_hidden_dim', kw.get('nabla_mu_hidden_dim', 0))
    d.mu_hidden_depth = kw.get('mu_hidden_depth', kw.get('nabla_mu_hidden_depth', 0))
    info = _lib.LayoutInfo()
    assert _lib.lib.rlrep_layout(C.byref(d), C.byref(info), None, 0) == 0, _lib.lib.rlrep_last_error()
    descs = (_lib.TensorDesc * info.n_tensors)()
    assert _lib.lib.rlrep_layout(C.byref(d), C.byref(info), descs, info.n_tensors) == 0
    mine = {t.name.decode(): (t.rows, t.cols) for t in descs if t.name.decode() != 'noise_alphabars'}
    ref = dict(param_shapes(alg, S, A, **kw))
    assert set(mine) == set(ref), set(mine) ^ set(ref)
    for k, shp in ref.items():
        r, c = mine[k]
        assert r * c == int(__import__('numpy').prod(shp)), k
    # no overlap, 16-byte aligned starts of every non-glued tensor, grads arena carries the reducible tail
    spans = sorted((t.arena, t.offset, t.offset + t.rows * t.cols) for t in descs)
    for (a0, s0, e0), (a1, s1, e1) in zip(spans, spans[1:]):
        assert a0 != a1 or e0 <= s1
    assert info.grad_floats == info.param_floats + _lib.GRAD_TAIL
    assert info.workspace_bytes > 0


def test_bad_arguments_are_rejected_with_a_message():
    from rlrep_amd import _lib
    d = _lib.Dims()
    info = _lib.LayoutInfo()
    assert _lib.lib.rlrep_layout(C.byref(d), C.byref(info), None, 0) < 0
    assert _lib.lib.rlrep_last_error()


def _plan(la, lb, R, Cn, K, lda=None, ldb=None, ldc=None):
    from rlrep_amd import _lib
    lda = lda if lda is not None else (R if la else K)
    ldb = ldb if ldb is not None else (Cn if lb else K)
    ldc = ldc if ldc is not None else Cn
    out = [C.c_int32() for _ in range(5)]
    assert _lib.lib.rlrep_gemm_plan(la, lb, R, Cn, K, lda, ldb, ldc, *[C.byref(o) for o in out]) == 0
    return dict(zip(('engine', 'tile', 'splits', 'kchunk', 'scalar'), (o.value for o in out)))


def test_gemm_routing_of_the_path_shapes(monkeypatch):
    """Host-only routing of the program builder (rlrep_gemm_plan): which engine / tile / split-K plan each layer class of
    the five agents gets.  The split plan must cover the inner dimension exactly once."""
    # the headline configuration's 256-wide layers stay on the latency-tuned 16-row engine
    assert _plan(0, 0, 256, 256, 256)['engine'] == 0
    assert _plan(0, 0, 256, 512, 256)['engine'] == 0
    assert _plan(1, 1, 256, 256, 256)['engine'] == 0
    # ctrlsac main.py dims: M = 256 layers (forward, dX) -> the 32 x 32 bf16x3 tile whose four waves split K (gemm_x3q_kernel): no slabs; with
    # RLREP_DISABLE=x3q the 64-wide tile (gemm_x3s_kernel) + split-K as before; weight gradients -> the 64-wide tile, no split
    for la, lb in ((0, 0), (0, 1)):
        p = _plan(la, lb, 256, 1024, 1024)
        assert (p['engine'], p['tile'], p['splits'], p['kchunk']) == (2, 32, 1, 1024), p
        assert _plan(la, lb, 256, 2048 if lb == 0 else 1024, 1024 if lb == 0 else 2048)['tile'] == 32      # phi.l3 forward / its dX
    assert _plan(0, 0, 256, 256, 2048)['tile'] == 64                   # the score matrix: 64 tiles of 32 x 32 are too few
    assert _plan(0, 0, 1024, 512, 512)['tile'] == 64                   # spedersac's M = 1024 critic layers stay on the 64-wide tile (R <= 256 only)
    monkeypatch.setenv('RLREP_DISABLE', 'x3q')
    p = _plan(0, 0, 256, 1024, 1024)
    assert (p['engine'], p['tile']) == (2, 64) and p['splits'] == 4 and p['splits'] * p['kchunk'] >= 1024 > (p['splits'] - 1) * p['kchunk']
    monkeypatch.delenv('RLREP_DISABLE')
    p = _plan(1, 1, 1024, 1024, 256)
    assert (p['engine'], p['tile'], p['splits']) == (2, 64, 1)
    # spedersac: both batches as one M = 2048 problem; the K = 119 first layer and its [512, 119] weight gradient have rows that are not
    # 16-byte regular ("scalar" sides): the 64-wide bf16x3 tile takes them through its any-alignment loaders
    assert _plan(0, 0, 2048, 512, 512)['engine'] == 2
    p = _plan(0, 0, 2048, 512, 119)
    assert (p['engine'], p['tile']) == (2, 64) and p['scalar'] == 3          # A and B rows of 119 floats
    p = _plan(1, 1, 512, 119, 2048, lda=512, ldb=119, ldc=119)
    assert (p['engine'], p['tile']) == (2, 64) and p['scalar'] == 6 and p['splits'] > 1
    # diffsrsac Humanoid nabla-mu head: all three passes on the bf16 pipe (dX / dW through transposed LDS reads), dX split along its long K; the
    # 256 x 128 persistent tile (tile = 256) wherever its tiles fill the 256 CUs evenly: 6 016 / 32 x 8 splits / 1 504 of them here
    assert _plan(0, 0, 2048, 96256, 512) == dict(engine=2, tile=256, splits=1, kchunk=512, scalar=0)
    p = _plan(0, 1, 2048, 512, 96256)
    assert p['engine'] == 2 and p['tile'] == 256 and p['splits'] == 8 and p['splits'] * p['kchunk'] >= 96256 > (p['splits'] - 1) * p['kchunk'] and p['kchunk'] % 32 == 0
    p = _plan(1, 1, 96256, 512, 2048)
    assert (p['engine'], p['tile'], p['splits']) == (2, 256, 1)
    # ... and not where they would leave most of a round of workgroups idle: 2048 x 2048 x 2048 = 128 of them
    assert _plan(0, 0, 2048, 2048, 2048)['tile'] == 128
    # RLREP_DISABLE=x3w keeps those products on the 128-wide tile (tests/test_gemm_engines.py holds both tiles to 2e-6 of float64)
    monkeypatch.setenv('RLREP_DISABLE', 'x3w')
    assert _plan(0, 0, 2048, 96256, 512)['tile'] == 128 and _plan(1, 1, 96256, 512, 2048)['tile'] == 128
    monkeypatch.delenv('RLREP_DISABLE')
    # every plan over a sweep: splits cover K, chunks are multiples of the 32-deep slice, tiles are 64, 128 or 256 (x 128)
    for R in (256, 1000, 2048, 4096):
        for Cn in (256, 520, 2048):
            for K in (64, 256, 1000, 4096, 50000):
                p = _plan(0, 0, R, Cn, K)
                if p['engine']:
                    assert p['tile'] in (32, 64, 128, 256) and (p['kchunk'] % 32 == 0 or p['tile'] == 32) and 1 <= p['splits'] <= 32
                    assert p['splits'] * p['kchunk'] >= K > (p['splits'] - 1) * p['kchunk'], (R, Cn, K, p)


def _nc_plan(heads, B, F, H):
    from rlrep_amd import _lib
    out = [C.c_int32() for _ in range(3)]
    assert _lib.lib.rlrep_nc_fwd_plan(heads, B, F, H, *[C.byref(o) for o in out]) == 0
    return tuple(o.value for o in out)


def test_noise_critic_engine_plan(monkeypatch):
    """Host-only: the vlsac noise critic's first layer goes to the bf16x3 engine when its K steps are 32 deep (8-row tiles, 128
    hidden units per workgroup once that still gives every CU a workgroup, else 64); other shapes stay on fp32 MFMA."""
    monkeypatch.delenv('RLREP_DISABLE', raising=False)
    assert _nc_plan(4, 256, 256, 256) == (1, 8, 128)       # headline critic step: target + live, two heads each
    assert _nc_plan(2, 256, 256, 256) == (1, 8, 64)        # actor step: the live heads only
    assert _nc_plan(2, 100, 96, 72)[0] == 1
    assert _nc_plan(2, 8, 8, 16)[0] == 0                   # tiny fixtures: K = 8
    assert _nc_plan(2, 256, 80, 256)[0] == 0               # 80 % 32 != 0
    monkeypatch.setenv('RLREP_DISABLE', 'x3')
    assert _nc_plan(4, 256, 256, 256)[0] == 0
