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
    assert _lib.lib.rlrep_abi_version() == 1


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
