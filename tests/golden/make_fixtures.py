#!/usr/bin/env python3
"""Generate golden vectors by importing the *reference* (/root/reference) in this container.

Runs ONLY where /root/reference exists (the build container).  Nothing here travels to the
GPU box except the .npz files it writes next to itself.  The reference sources are imported,
never copied: what is stored is data (inputs, recorded noise/indices, outputs).

Recipe:
  * sys.dont_write_bytecode (the reference mount is read-only),
  * stub modules `gym` and `torchinfo` (only their *names* are needed by the agents),
  * torch.set_num_threads(1) so the fp32 results are run-to-run bit-identical,
  * record every RNG draw the hot path makes (Appendix B of SURVEY.md):
      np.random.randint                         -> replay indices
      torch.distributions.normal._standard_normal -> Normal.rsample eps
      torch.randint / torch.normal              -> diffsrsac noise index / perturbation
  * hook every optimizer.step() to snapshot the gradients it consumes.

Fixture layout (flat npz, '/'-separated keys):
  meta/json                      json string: alg, dims, ctor kwargs, T, batch size, feature iters
  replay/{state,action,next_state,reward,done}   the synthetic replay content (float32)
  init/<module>.<param>          initial parameters (float32; log_alpha float64; vlsac critic.noise)
  t<k>/idx/<i>                   i-th np.random.randint draw of train() call k
  t<k>/eps/<i>                   i-th torch noise draw of train() call k (in draw order)
  t<k>/info/<key>                returned metrics of train() call k (float64)
  t<k>/grad/<opt>#<j>/<name>     gradient seen by the j-th .step() of optimizer <opt> in call k
                                  (full tensors for the 'tiny' fixtures, [l2norm,sum,first16] for 'cfg')
  final/<module>.<param>         parameters after T train() calls (same full/summary policy)
  adam/<opt>/<name>/{m,v,step}   Adam state after T calls (summary policy always)

Usage: python tests/golden/make_fixtures.py [--only vlsac_tiny ...]
"""
import sys
sys.dont_write_bytecode = True
import os, types, json, argparse
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    if not os.path.isdir(REF):
        raise SystemExit('reference not present; fixtures can only be generated in the build container')
    sys.modules.setdefault('gym', types.ModuleType('gym'))
    ti = types.ModuleType('torchinfo')
    ti.summary = lambda *a, **k: None
    sys.modules.setdefault('torchinfo', ti)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from utils import buffer as rbuffer            # noqa
    from agent.sac import sac_agent                # noqa
    from agent.vlsac import vlsac_agent            # noqa
    from agent.ctrlsac import ctrlsac_agent        # noqa
    from agent.spedersac import spedersac_agent    # noqa
    from agent.diffsrsac import diffsrsac_agent    # noqa
    return dict(buffer=rbuffer, sac=sac_agent, vlsac=vlsac_agent, ctrlsac=ctrlsac_agent,
                spedersac=spedersac_agent, diffsrsac=diffsrsac_agent)


class ActionSpace:
    def __init__(self, dim, bound):
        self.low = -bound * np.ones(dim, dtype=np.float32)
        self.high = bound * np.ones(dim, dtype=np.float32)


class Recorder:
    """Monkeypatch the RNG entry points of the hot path with recording wrappers."""

    def __init__(self, source=None):
        # source=None: record what the reference's own generators draw (tiny fixtures).
        # source=NoiseSource: SUPPLY deterministic numpy draws instead (config-dims fixtures), so the
        # noise does not have to be stored.
        self.idx, self.eps = [], []
        self.source = source
        import torch.distributions.normal as tdn
        self._tdn = tdn
        self._o_sn = tdn._standard_normal
        self._o_ri = np.random.randint
        self._o_tri = torch.randint
        self._o_tn = torch.normal

        def sn(shape, dtype, device):
            if self.source is not None:
                e = torch.from_numpy(self.source.normal(tuple(shape))).to(dtype)
            else:
                e = self._o_sn(shape, dtype, device)
            self.eps.append(e.detach().clone().numpy())
            return e

        def ri(low, high=None, size=None, **k):
            if self.source is not None:
                assert low == 0
                r = self.source.indices(high, size)
            else:
                r = self._o_ri(low, high, size=size, **k)
            self.idx.append(np.asarray(r).copy())
            return r

        def tri(low, high, size, **k):
            if self.source is not None:
                assert low == 0
                r = torch.from_numpy(self.source.indices(high, size[0]))
            else:
                r = self._o_tri(low, high, size, **k)
            self.eps.append(r.detach().clone().numpy())
            return r

        def tn(mean, std, **k):
            if self.source is not None:
                r = mean + std * torch.from_numpy(self.source.normal(tuple(mean.shape)))
            else:
                r = self._o_tn(mean, std, **k)
            self.eps.append(r.detach().clone().numpy())
            return r

        tdn._standard_normal = sn
        np.random.randint = ri
        torch.randint = tri
        torch.normal = tn

    def take(self):
        i, e = self.idx, self.eps
        self.idx, self.eps = [], []
        return i, e

    def close(self):
        self._tdn._standard_normal = self._o_sn
        np.random.randint = self._o_ri
        torch.randint = self._o_tri
        torch.normal = self._o_tn


def summary(a):
    a = np.asarray(a, dtype=np.float64).ravel()
    head = np.zeros(16)
    head[:min(16, a.size)] = a[:16]
    return np.concatenate([[np.sqrt((a * a).sum()), a.sum()], head])


MODULES = {
    'sac': ['critic', 'critic_target', 'actor'],
    'vlsac': ['critic', 'critic_target', 'actor', 'encoder', 'decoder', 'f', 'f_target'],
    'ctrlsac': ['critic', 'critic_target', 'actor', 'phi', 'phi_target', 'mu', 'theta',
                'frozen_phi', 'frozen_phi_target'],
    'spedersac': ['critic', 'critic_target', 'actor', 'phi', 'phi_target', 'mu', 'theta'],
    'diffsrsac': ['critic', 'critic_target', 'actor', 'critic_feed_feature', 'nablamu_net'],
}
OPTS = {
    'sac': ['critic_optimizer', 'actor_optimizer', 'log_alpha_optimizer'],
    'vlsac': ['feature_optimizer', 'critic_optimizer', 'actor_optimizer', 'log_alpha_optimizer'],
    'ctrlsac': ['feature_optimizer', 'critic_optimizer', 'actor_optimizer', 'log_alpha_optimizer'],
    'spedersac': ['feature_optimizer', 'critic_optimizer', 'actor_optimizer', 'log_alpha_optimizer'],
    'diffsrsac': ['phi_optimizer', 'nablamu_net_optimizer', 'critic_optimizer', 'actor_optimizer',
                  'log_alpha_optimizer'],
}


def named_state(agent, alg):
    out = {}
    for m in MODULES[alg]:
        if not hasattr(agent, m):
            continue                      # use_feature_target=False: no f_target / phi_target / frozen_phi_target attributes
        for k, v in getattr(agent, m).state_dict().items():
            out[f'{m}.{k}'] = v.detach().clone().numpy()
    out['log_alpha'] = agent.log_alpha.detach().clone().numpy()
    if alg == 'vlsac':
        out['critic.noise'] = agent.critic.noise.detach().clone().numpy()
        out['critic_target.noise'] = agent.critic_target.noise.detach().clone().numpy()
    if alg == 'diffsrsac':
        out['noise_alphabars'] = agent.noise_alphabars.detach().clone().numpy()
    return out


def param_names(agent, alg):
    """id(param) -> '<module>.<name>' for every live parameter (+ log_alpha)."""
    names = {id(agent.log_alpha): 'log_alpha'}
    for m in MODULES[alg]:
        if not hasattr(agent, m):
            continue
        for k, p in getattr(agent, m).named_parameters():
            names[id(p)] = f'{m}.{k}'
    return names


def synth_replay(mods, S, A, n):
    import synth
    data = synth.replay(S, A, n)
    buf = mods['buffer'].ReplayBuffer(S, A, max_size=n)
    for k, v in data.items():
        getattr(buf, k)[:] = v
    buf.size = n
    buf.ptr = 0
    return buf


def load_synth_init(agent, alg, S_A, kw_shapes):
    """Config-dims fixtures: overwrite every parameter with synth.init_like values (regenerable on the
    GPU box), then re-tie the targets exactly as the reference constructors do (hard copies)."""
    import synth
    shapes = []
    for m in MODULES[alg]:
        for k, v in getattr(agent, m).state_dict().items():
            shapes.append((f'{m}.{k}', tuple(v.shape)))
    if alg == 'vlsac':
        shapes.append(('critic.noise', tuple(agent.critic.noise.shape)))
    # the travelling shape table must reproduce the reference's state_dict order exactly
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle.shapes import param_shapes
    mine = param_shapes(alg, S_A[0], S_A[1], **kw_shapes)
    assert [(n, tuple(s)) for n, s in mine] == shapes, 'oracle/shapes.py disagrees with the reference'
    vals = synth.init_like(shapes)
    with torch.no_grad():
        for m in MODULES[alg]:
            sd = {k: torch.from_numpy(vals[f'{m}.{k}']) for k in getattr(agent, m).state_dict().keys()}
            getattr(agent, m).load_state_dict(sd)
        agent.critic_target.load_state_dict(agent.critic.state_dict())
        if alg == 'vlsac':
            agent.f_target.load_state_dict(agent.f.state_dict())
            agent.critic.noise = torch.from_numpy(vals['critic.noise']).clone()
            agent.critic_target.noise = torch.from_numpy(vals['critic.noise']).clone()
        if alg in ('ctrlsac', 'spedersac'):
            agent.phi_target.load_state_dict(agent.phi.state_dict())


def make(mods, name, alg, S, A, bound, B, T, kwargs, full, replay_n=256, patch_vae_hidden=None, keep_grads=True):
    import synth
    torch.set_num_threads(1)
    torch.manual_seed(0)
    np.random.seed(0)
    saved_defaults = None
    if patch_vae_hidden is not None:
        # The reference hard-wires hidden_dim=256 for Encoder/Decoder/GaussianFeature through ctor
        # *defaults* (networks/vae.py:24,70,101).  For the tiny fixture only, the default value is
        # overridden at run time (no reference source is edited) so the vectors stay small.
        import networks.vae as rvae
        saved_defaults = {}
        for cls in (rvae.Encoder, rvae.Decoder, rvae.GaussianFeature):
            saved_defaults[cls] = cls.__init__.__defaults__
            d = list(cls.__init__.__defaults__)
            d[-1] = patch_vae_hidden          # hidden_dim is the last defaulted arg in all three
            cls.__init__.__defaults__ = tuple(d)
    try:
        cls = {'sac': mods['sac'].SACAgent, 'vlsac': mods['vlsac'].VLSACAgent,
               'ctrlsac': mods['ctrlsac'].CTRLSACAgent, 'spedersac': mods['spedersac'].SPEDERSACAgent,
               'diffsrsac': mods['diffsrsac'].DIFFSRSACAgent}[alg]
        agent = cls(state_dim=S, action_dim=A, action_space=ActionSpace(A, bound), **kwargs)
    finally:
        if saved_defaults:
            for c, d in saved_defaults.items():
                c.__init__.__defaults__ = d
    buf = synth_replay(mods, S, A, replay_n)
    out = {}
    if not full:
        load_synth_init(agent, alg, (S, A), kwargs)
    if full:
        # tiny fixtures carry their inputs; config-dims fixtures regenerate them from tests/golden/synth.py
        for k in ('state', 'action', 'next_state', 'reward', 'done'):
            out[f'replay/{k}'] = getattr(buf, k).astype(np.float32)
        for k, v in named_state(agent, alg).items():
            out[f'init/{k}'] = v
    elif alg == 'diffsrsac':
        out['init/noise_alphabars'] = agent.noise_alphabars.numpy()
    names = param_names(agent, alg)

    # hook optimizer steps
    grads_log = []
    counters = {}

    def hook(optname, opt):
        orig = opt.step

        def step(*a, **k):
            j = counters.get(optname, 0)
            counters[optname] = j + 1
            for g in opt.param_groups:
                for p in g['params']:
                    if p.grad is not None:
                        grads_log.append((f'{optname}#{j}', names.get(id(p), '?'), p.grad.detach().clone().numpy()))
            return orig(*a, **k)
        opt.step = step
    for o in OPTS[alg]:
        hook(o, getattr(agent, o))

    rec = Recorder(source=None if full else synth.NoiseSource())
    try:
        for t in range(T):
            counters.clear()
            grads_log.clear()
            info = agent.train(buf, B)
            idx, eps = rec.take()
            if full:
                for i, a in enumerate(idx):
                    out[f't{t}/idx/{i}'] = a.astype(np.int64)
                for i, e in enumerate(eps):
                    out[f't{t}/eps/{i}'] = e
            for k, v in info.items():
                out[f't{t}/info/{k}'] = np.float64(v.item() if torch.is_tensor(v) else v)
            for (o, n, g) in (grads_log if keep_grads else []):
                out[f't{t}/grad/{o}/{n}'] = g if full else summary(g)
    finally:
        rec.close()
    for k, v in named_state(agent, alg).items():
        out[f'final/{k}'] = v if full else summary(v)
    for o in OPTS[alg]:
        opt = getattr(agent, o)
        for g in opt.param_groups:
            for p in g['params']:
                st = opt.state.get(p, None)
                if not st:
                    continue
                n = names.get(id(p), '?')
                out[f'adam/{o}/{n}/m'] = summary(st['exp_avg'].numpy())
                out[f'adam/{o}/{n}/v'] = summary(st['exp_avg_sq'].numpy())
                out[f'adam/{o}/{n}/step'] = np.float64(float(st['step']))
    meta = dict(name=name, alg=alg, S=S, A=A, bound=bound, B=B, T=T, kwargs=kwargs, full=full,
                replay_n=replay_n, patch_vae_hidden=patch_vae_hidden,
                torch=torch.__version__, numpy=np.__version__, threads=1,
                steps=agent.steps)
    out['meta/json'] = np.array(json.dumps(meta))
    path = os.path.join(HERE, f'{name}.npz')
    np.savez_compressed(path, **out)
    print(f'{name}: {len(out)} arrays, {os.path.getsize(path)/1e6:.2f} MB')


SPED = dict(phi_and_mu_lr=1e-5, phi_hidden_depth=1, mu_hidden_depth=0, critic_and_actor_lr=3e-4,
            extra_feature_steps=5)

CASES = {
    # name: (alg, S, A, bound, B, T, kwargs, full, replay_n, patch_vae_hidden)
    'sac_tiny': ('sac', 5, 3, 1.0, 8, 3, dict(hidden_dim=16), True, 64, None),
    'sac_pendulum': ('sac', 3, 1, 2.0, 64, 2, dict(hidden_dim=256), False, 256, None),
    'vlsac_tiny': ('vlsac', 5, 3, 1.0, 8, 3, dict(hidden_dim=16, feature_dim=8, extra_feature_steps=3), True, 64, 16),
    'vlsac_hc': ('vlsac', 17, 6, 1.0, 256, 2, dict(hidden_dim=256, feature_dim=256, extra_feature_steps=3), False, 1024, None),
    'ctrlsac_tiny': ('ctrlsac', 5, 3, 1.0, 8, 3, dict(hidden_dim=16, feature_dim=8, extra_feature_steps=3), True, 64, None),
    'ctrlsac_hc256': ('ctrlsac', 17, 6, 1.0, 256, 2, dict(hidden_dim=256, feature_dim=256, extra_feature_steps=3), False, 1024, None),
    'spedersac_tiny': ('spedersac', 5, 3, 1.0, 8, 3, dict(SPED, phi_hidden_dim=16, mu_hidden_dim=16, critic_and_actor_hidden_dim=16, feature_dim=8, hidden_dim=16), True, 64, None),
    'spedersac_ant512': ('spedersac', 111, 8, 1.0, 1024, 1, dict(SPED, phi_hidden_dim=512, mu_hidden_dim=512, critic_and_actor_hidden_dim=256, feature_dim=512, hidden_dim=256), False, 4096, None),
    'diffsrsac_tiny': ('diffsrsac', 5, 3, 1.0, 8, 3, dict(feature_dim=8, phi_hidden_dim=16, nabla_mu_hidden_dim=16, hidden_dim=16, extra_feature_steps=3), True, 64, None),
    'diffsrsac_hc': ('diffsrsac', 17, 6, 1.0, 256, 1, dict(hidden_dim=256, extra_feature_steps=3), False, 1024, None),
    # BASELINE configs 3 and 5 at the dimensions main.py runs them with (main.py:90-91 ctrlsac F=2048 / H=1024; Humanoid-v3 S=376 A=17,
    # batch 2048); summaries only (24 M / 50 M parameters)
    'ctrlsac_hc2048': ('ctrlsac', 17, 6, 1.0, 256, 2, dict(hidden_dim=1024, feature_dim=2048, extra_feature_steps=3), False, 1024, None),
    'diffsrsac_humanoid_b2048': ('diffsrsac', 376, 17, 0.4, 2048, 1, dict(hidden_dim=256, extra_feature_steps=3), False, 4096, None),
    # 25 consecutive train() calls at the headline dimensions (SURVEY.md 7.4 / 8(c)(v) free-run loss trace): metrics of every call,
    # final parameters and Adam state; per-step gradients are not stored
    # use_feature_target=False (vlsac_agent.py:176-179,214-219; ctrlsac_agent.py:268-273,340-346; spedersac_agent.py:306-307)
    'vlsac_tiny_noft': ('vlsac', 5, 3, 1.0, 8, 3, dict(hidden_dim=16, feature_dim=8, extra_feature_steps=3, use_feature_target=False), True, 64, 16),
    'ctrlsac_tiny_noft': ('ctrlsac', 5, 3, 1.0, 8, 2, dict(hidden_dim=16, feature_dim=8, extra_feature_steps=3, use_feature_target=False), True, 64, None),
    'spedersac_tiny_noft': ('spedersac', 5, 3, 1.0, 8, 2, dict(SPED, phi_hidden_dim=16, mu_hidden_dim=16, critic_and_actor_hidden_dim=16, feature_dim=8, hidden_dim=16, use_feature_target=False), True, 64, None),
    # ELU-layer regulariser switched on (diffsrsac_agent.py:62-90,215-227): enters q_loss_reg only
    'diffsrsac_tiny_reg': ('diffsrsac', 5, 3, 1.0, 8, 2, dict(feature_dim=8, phi_hidden_dim=16, nabla_mu_hidden_dim=16, hidden_dim=16, extra_feature_steps=3, critic_elu_layer_regularizer_lambda=0.25), True, 64, None),
    'vlsac_hc_free25': ('vlsac', 17, 6, 1.0, 256, 25, dict(hidden_dim=256, feature_dim=256, extra_feature_steps=3), False, 1024, None, False),
}


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', nargs='*', default=None)
    args = ap.parse_args()
    mods = _import_reference()
    for name, c in CASES.items():
        if args.only and name not in args.only:
            continue
        make(mods, name, *c[:7], c[7], c[8], c[9], *c[10:])
