"""GPU parity at the LARGE dimensions of BASELINE configs 3-5, where the step programs route their GEMMs to the
LDS-tiled engine (gemm_lds.hip: fp32-MFMA 64/128 tiles, split-K) and, for diffsrsac's nabla-mu head, to its bf16x3
variant.  No golden fixture exists at these sizes (the parameters alone are 25-200 MB), so the check is HIP vs the CPU
oracle on synthetic parameters / replay / injected noise, at sizes the oracle finishes in seconds.

Tolerance: BASELINE.json's 1e-4 relative (metrics: |x-ref| <= 1e-4*max(|ref|,1e-2); parameters: relative L2 per tensor)."""
import os
import numpy as np
import pytest
import torch

from fixture_io import rel_l2      # puts tests/golden on sys.path
import synth

pytestmark = pytest.mark.gpu


class _Space:
    def __init__(self, A):
        self.low, self.high = -np.ones(A, np.float32), np.ones(A, np.float32)


def _retie(alg, P):
    ties = [('critic', 'critic_target')]
    if alg in ('ctrlsac', 'spedersac'):
        ties.append(('phi', 'phi_target'))
    for s, d in ties:
        for k in list(P):
            if k.startswith(s + '.') and (d + k[len(s):]) in P:
                P[d + k[len(s):]] = P[k].copy()
    if alg == 'ctrlsac':
        for d in ('frozen_phi', 'frozen_phi_target'):
            for k in list(P):
                if k.startswith('phi.') and (d + k[3:]) in P:
                    P[d + k[3:]] = P[k].copy()


def _run(alg, cls_path, S, A, B, kw, trains=1, replay_n=4096, expect_kernels=(), agent_kw=None):
    import importlib
    from oracle import make_oracle
    from oracle.agents import gather_batch
    from oracle.shapes import param_shapes
    mod, name = cls_path
    cls = getattr(importlib.import_module(mod), name)
    init = synth.init_like(param_shapes(alg, S, A, **kw), seed=99)
    _retie(alg, init)
    init['log_alpha'] = np.log(np.float64(0.1))
    if alg == 'vlsac':
        init['critic.noise'] = np.random.RandomState(5).standard_normal(init['critic.noise'].shape).astype(np.float32)   # N(0,1) buffer
        init['critic_target.noise'] = init['critic.noise'].copy()                       # quirk Q3: same buffer in both copies
    agent = cls(state_dim=S, action_dim=A, action_space=_Space(A), max_batch=B, graph=False, **{k: v for k, v in kw.items() if k != 'vae_hidden'}, **(agent_kw or {}))
    if alg == 'diffsrsac':
        init['noise_alphabars'] = agent.core.state()['noise_alphabars'].numpy().copy()
    agent.core.load_state(init)
    names = [agent.core.stages(p) for p in range(7)]
    flat = ' | '.join(n for st in names for n in st)
    from rlrep_amd.utils.buffer import ReplayBuffer
    data = synth.replay(S, A, replay_n, seed=3)
    buf = ReplayBuffer(S, A, max_size=replay_n)
    buf.load(data['state'], data['action'], data['next_state'], data['reward'], data['done'])
    o = make_oracle(alg, S, A, init, **kw)
    if alg == 'diffsrsac':
        o.P['noise_alphabars'] = torch.from_numpy(init['noise_alphabars'])
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    rs = np.random.RandomState(11)
    nf = kw.get('extra_feature_steps', 0) + 1
    tens = {k: torch.from_numpy(v) for k, v in data.items()}
    for t in range(trains):
        idx = [rs.randint(0, replay_n, size=B) for _ in range(o.n_batches())]
        eps = []
        if alg == 'vlsac':
            eps = [rs.standard_normal((B, kw['feature_dim'])).astype(np.float32) for _ in range(nf)]
        if alg == 'diffsrsac':
            for _ in range(nf):
                eps += [rs.randint(0, 1000, size=B), (0.449 * rs.standard_normal((B, S))).astype(np.float32)]
        eps += [rs.standard_normal((B, A)).astype(np.float32) for _ in range(2)]
        info = agent.train_injected(buf, B, idx, eps)
        oinfo = o.train([gather_batch(tens, i) for i in idx], [torch.as_tensor(e) for e in eps])
        for k, v in oinfo.items():
            assert np.isfinite(float(info[k])), (alg, k)
            assert abs(info[k] - v) <= 1e-4 * max(abs(v), 1e-2), (alg, t, k, info[k], v)
    st, P = agent.core.state(), o.state()
    worst = 0.0
    for k in st:
        if k in P and not k.endswith('noise') and k != 'noise_alphabars':
            e = rel_l2(st[k].numpy(), P[k].numpy())
            worst = max(worst, e)
            assert e < 1e-4, (alg, k, e)
    print(f'{alg} S={S} B={B}: worst parameter rel-L2 after {trains} train(): {worst:.2e}')
    return agent


def test_ctrlsac_main_py_dimensions():
    """ctrlsac as main.py:87-91 builds it: feature_dim 2048, hidden 1024, B = 256 (M = 256 layers: 64-wide tiles + split-K)"""
    _run('ctrlsac', ('rlrep_amd.agent.ctrlsac.ctrlsac_agent', 'CTRLSACAgent'), 17, 6, 256,
         dict(hidden_dim=1024, feature_dim=2048, extra_feature_steps=1))


def test_spedersac_ant_dimensions_two_trains():
    _run('spedersac', ('rlrep_amd.agent.spedersac.spedersac_agent', 'SPEDERSACAgent'), 111, 8, 1024,
         dict(phi_and_mu_lr=1e-5, phi_hidden_dim=512, phi_hidden_depth=1, mu_hidden_dim=512, mu_hidden_depth=0,
              critic_and_actor_lr=3e-4, critic_and_actor_hidden_dim=256, feature_dim=512, hidden_dim=256, extra_feature_steps=1),
         trains=2)


def test_spedersac_split_k_gradients_summed_by_the_optimizer_launch(monkeypatch):
    """Builder::fold_fin (AdamTask::Slab with padded slab rows): the split-K partials of phi / mu's weight gradients -- [512, 119] and [512, 111]
    among them, slab rows padded to 120 / 112 floats -- are added in split order by the feature group's optimizer launch instead of a finishing
    launch.  Against RLREP_DISABLE=fold_dwfin (a finishing launch: one more per feature step) and RLREP_ENABLE=fin_inline on top of it (the 64-wide
    bf16x3 tile finishes EVERY split-K product inside the launch -- the last split workgroup of a tile, FLAG_FIN_INLINE; opt-in: measured slower):
    parameters and moments BIT-identical after two train() calls at Ant dimensions (all forms are checked against the oracle by _run)."""
    outs, counts = [], []
    for off, on in (('', ''), ('fold_dwfin', ''), ('fold_dwfin', 'fin_inline')):
        if off:
            monkeypatch.setenv('RLREP_DISABLE', off)
        if on:
            monkeypatch.setenv('RLREP_ENABLE', on)
        from rlrep_amd import _lib
        n0 = _lib.lib.rlrep_launch_counter()
        a = _run('spedersac', ('rlrep_amd.agent.spedersac.spedersac_agent', 'SPEDERSACAgent'), 111, 8, 1024,
                 dict(phi_and_mu_lr=1e-5, phi_hidden_dim=512, phi_hidden_depth=1, mu_hidden_dim=512, mu_hidden_depth=0,
                      critic_and_actor_lr=3e-4, critic_and_actor_hidden_dim=256, feature_dim=512, hidden_dim=256, extra_feature_steps=1),
                 trains=2)
        st = {k: v.numpy().copy() for k, v in a.core.state().items()}
        st['exp_avg'] = a.core.exp_avg.cpu().numpy().copy(); st['exp_avg_sq'] = a.core.exp_avg_sq.cpu().numpy().copy()
        outs.append(st)
        counts.append(_lib.lib.rlrep_launch_counter() - n0)
        del a
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), k
        assert np.array_equal(outs[0][k], outs[2][k]), k
    assert counts[1] == counts[0] + 2 * 2, counts          # (two train() calls x two feature steps)
    assert counts[2] < counts[0], counts                   # (no finishing launch behind any split stage)


def test_diffsrsac_wide_nabla_mu_head_on_bf16x3():
    """S = 76, F = 256, B = 1024: the nabla-mu head is 1024 x 512 x 19 456 (20 GFLOP per pass) -- the size class whose
    forward and dX the builder sends to the bf16x3 tile (Humanoid: 2048 x 512 x 96 256)"""
    _run('diffsrsac', ('rlrep_amd.agent.diffsrsac.diffsrsac_agent', 'DIFFSRSACAgent'), 76, 8, 1024,
         dict(hidden_dim=256, extra_feature_steps=1))


def test_large_engines_off_gives_the_same_step(monkeypatch):
    """RLREP_DISABLE=gemm_lds keeps every GEMM on the 16-row engine: same train() within fp32 summation-order noise"""
    outs = []
    for off in (False, True):
        if off:
            monkeypatch.setenv('RLREP_DISABLE', 'gemm_lds')
        a = _run('ctrlsac', ('rlrep_amd.agent.ctrlsac.ctrlsac_agent', 'CTRLSACAgent'), 17, 6, 128,
                 dict(hidden_dim=512, feature_dim=1024, extra_feature_steps=0))
        outs.append({k: v.numpy().copy() for k, v in a.core.state().items()})
    for k in outs[0]:
        assert rel_l2(outs[0][k], outs[1][k]) < 1e-5, k


@pytest.mark.parametrize('S,A,B,F,H', [(17, 6, 256, 256, 256), (11, 3, 100, 96, 72), (9, 2, 37, 64, 40), (11, 3, 50, 96, 96),
                                       (5, 2, 5, 64, 32), (9, 4, 260, 64, 32)])
def test_vlsac_noise_critic_first_layer_on_bf16x3(S, A, B, F, H, monkeypatch):
    """vlsac at dimensions whose noise-critic first layer (vlsac_agent.py:44-63; 63 % of a train()'s FLOPs) runs on the bf16x3
    engine (noisecritic.hip nc_fwd_x3_kernel): the headline shape, and ragged ones -- batch not a multiple of the 8-row tile,
    hidden width not a multiple of 16, three and two K steps -- against the CPU oracle, two train() calls each.  The last case
    (H = 96) also takes the bf16x3 form of the dX launch (nc_dx_x3_kernel: H % 32 == 0) with a batch that is not a multiple of its
    4-row tile and F = 96 = one and a half of its 64-column tiles; the weight gradient is the bf16x3 split-K kernel in every case
    (nc_dw_x3_kernel: ragged last split, partial 64-wide tiles).  The last two: a batch smaller than one tile, and one split of 260 rows
    over 8 ranges."""
    import ctypes as C
    from rlrep_amd import _lib
    monkeypatch.delenv('RLREP_DISABLE', raising=False)
    out = [C.c_int32() for _ in range(3)]
    assert _lib.lib.rlrep_nc_fwd_plan(2, B, F, H, *[C.byref(o) for o in out]) == 0 and out[0].value == 1
    _run('vlsac', ('rlrep_amd.agent.vlsac.vlsac_agent', 'VLSACAgent'), S, A, B,
         dict(hidden_dim=H, feature_dim=F, extra_feature_steps=1), trains=2)


@pytest.mark.parametrize('S,A,B,F,H,Hv', [(7, 2, 19, 24, 32, 40), (5, 3, 33, 40, 16, 24), (9, 2, 64, 16, 32, 72)])
def test_vlsac_fused_heads_and_vae_mid_at_odd_shapes(S, A, B, F, H, Hv):
    """heads_vae_kernel (the Gaussian heads of encoder / f fused with vae_mid; csrc/elementwise.hip) at shapes where nothing is a multiple
    of its 16 x 16 tile or its 16-deep chunks: feature width 24 / 40 / 16, batch 19 / 33 / 64, VAE hidden width 40 / 24 / 72 -- against
    the CPU oracle, two train() calls each."""
    _run('vlsac', ('rlrep_amd.agent.vlsac.vlsac_agent', 'VLSACAgent'), S, A, B,
         dict(hidden_dim=H, feature_dim=F, vae_hidden=Hv, extra_feature_steps=1), trains=2, agent_kw=dict(vae_hidden_dim=Hv))


def test_vlsac_noise_critic_engines_agree(monkeypatch):
    """RLREP_DISABLE=x3 keeps the first layer on fp32 MFMA: the same two train() calls end within fp32 rounding of the bf16x3 run."""
    outs = []
    for off in (False, True):
        if off:
            monkeypatch.setenv('RLREP_DISABLE', 'x3')
        else:
            monkeypatch.delenv('RLREP_DISABLE', raising=False)
        a = _run('vlsac', ('rlrep_amd.agent.vlsac.vlsac_agent', 'VLSACAgent'), 17, 6, 128,
                 dict(hidden_dim=128, feature_dim=128, extra_feature_steps=0), trains=2)
        outs.append({k: v.numpy().copy() for k, v in a.core.state().items()})
    for k in outs[0]:
        assert rel_l2(outs[0][k], outs[1][k]) < 1e-5, k


def test_diffsrsac_regulariser_at_config_dimensions():
    """critic_elu_layer_regularizer_lambda != 0 (diffsrsac_agent.py:62-90, 215-227) at B = 1024, H = 256: the Gram matrices x^T x
    [256 x 256] reduce over 1024 rows through the LDS-tiled weight-gradient engine (split-K slabs reserved whatever lambda is)."""
    _run('diffsrsac', ('rlrep_amd.agent.diffsrsac.diffsrsac_agent', 'DIFFSRSACAgent'), 17, 6, 1024,
         dict(hidden_dim=256, extra_feature_steps=0, critic_elu_layer_regularizer_lambda=0.5), trains=2)
