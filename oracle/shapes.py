"""Ordered (name, shape) lists of every agent's parameters (oracle; test infrastructure only).

Names and order reproduce `state_dict()` of the reference modules, module by module in the order
`tests/golden/make_fixtures.py::MODULES` lists them (checked there against the imported reference),
so that `tests/golden/synth.py::init_like` regenerates the config-dims fixtures' initial parameters.
"""


def _lin(name, out_f, in_f):
    return [(name + '.weight', (out_f, in_f)), (name + '.bias', (out_f,))]


def _mlp(prefix, in_f, hid, out_f, depth):
    """utils/util.py:85-96 Sequential indexing: Linear at 0,2,4,..."""
    if depth == 0:
        return _lin(f'{prefix}.0', out_f, in_f)
    out = _lin(f'{prefix}.0', hid, in_f)
    for i in range(1, depth):
        out += _lin(f'{prefix}.{2 * i}', hid, hid)
    out += _lin(f'{prefix}.{2 * depth}', out_f, hid)
    return out


def _actor(S, A, H):
    return _mlp('actor.trunk', S, H, 2 * A, 2)            # agent/sac/actor.py:63-74


def _six(prefix, in_f, H):
    """l1..l6 critics (vlsac_agent.py:33-41, spedersac_agent.py:26-34, diffsrsac_agent.py:51-59)."""
    return (_lin(prefix + '.l1', H, in_f) + _lin(prefix + '.l2', H, H) + _lin(prefix + '.l3', 1, H) +
            _lin(prefix + '.l4', H, in_f) + _lin(prefix + '.l5', H, H) + _lin(prefix + '.l6', 1, H))


def _gauss(prefix, in_f, Hv, F):
    return (_lin(prefix + '.l1', Hv, in_f) + _lin(prefix + '.l2', Hv, Hv) +
            _lin(prefix + '.mean_linear', F, Hv) + _lin(prefix + '.log_std_linear', F, Hv))


def param_shapes(alg, S, A, **kw):
    H = kw.get('hidden_dim', 256)
    if alg == 'sac':
        out = []
        for m in ('critic', 'critic_target'):
            out += _mlp(m + '.Q1', S + A, H, 1, 2) + _mlp(m + '.Q2', S + A, H, 1, 2)
        return out + _actor(S, A, H)
    if alg == 'vlsac':
        F = kw.get('feature_dim', 256)
        Hv = kw.get('vae_hidden', 256)                    # networks/vae.py ctor default
        out = _six('critic', F, H) + _six('critic_target', F, H) + _actor(S, A, H)
        out += _gauss('encoder', 2 * S + A, Hv, F)
        out += _lin('decoder.l1', Hv, F) + _lin('decoder.state_linear', S, Hv) + _lin('decoder.reward_linear', 1, Hv)
        out += _gauss('f', S + A, Hv, F) + _gauss('f_target', S + A, Hv, F)
        out += [('critic.noise', (20, F))]
        return out
    if alg == 'ctrlsac':
        F = kw.get('feature_dim', 2048)
        out = []
        for m in ('critic', 'critic_target'):
            out += _lin(m + '.l1', H, F) + _lin(m + '.l2', 1, H) + _lin(m + '.l4', H, F) + _lin(m + '.l5', 1, H)
        out += _actor(S, A, 256)                           # ctrlsac_agent.py:188-194 hard-codes 256
        phi = lambda p: _lin(p + '.l1', H, S + A) + _lin(p + '.l2', H, H) + _lin(p + '.l3', F, H)
        out += phi('phi') + phi('phi_target')
        out += _lin('mu.l1', H, S) + _lin('mu.l2', H, H) + _lin('mu.l3', F, H)
        out += _lin('theta.l', 1, F)
        out += phi('frozen_phi') + phi('frozen_phi_target')
        return out
    if alg == 'spedersac':
        F = kw.get('feature_dim', 2048)
        Hc = kw['critic_and_actor_hidden_dim']
        out = _six('critic', F, Hc) + _six('critic_target', F, Hc) + _actor(S, A, Hc)
        pd, md = kw['phi_hidden_depth'], kw['mu_hidden_depth']
        out += _mlp('phi.trunk', S + A, kw['phi_hidden_dim'], F, pd)
        out += _mlp('phi_target.trunk', S + A, kw['phi_hidden_dim'], F, pd)
        out += _mlp('mu.trunk', S, kw['mu_hidden_dim'], F, md)
        out += _lin('theta.l', 1, F)
        return out
    if alg == 'diffsrsac':
        F = kw.get('feature_dim', 256)
        out = _six('critic', F, H) + _six('critic_target', F, H) + _actor(S, A, H)
        out += _mlp('critic_feed_feature.z_vector', S + A, kw.get('phi_hidden_dim', 256), F,
                    kw.get('phi_hidden_depth', 1))
        out += _mlp('nablamu_net.Mu_z_by_s_layer', S + 1, kw.get('nabla_mu_hidden_dim', 512), F * S,
                    kw.get('nabla_mu_hidden_depth', 1))
        return out
    raise ValueError(alg)
