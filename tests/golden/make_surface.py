#!/usr/bin/env python3
"""Capture the reference's CALL SURFACE and module known-answers by importing /root/reference (build container only).

Writes, next to itself under surface/:
  signatures.json   inspect.signature of every class constructor and public method the drop-in boundary mirrors
                    (agents, ReplayBuffer, util helpers, networks/{vae,critic,policy}.py, agent/sac/{actor,critic}.py)
  modules.npz       for every nn.Module class of those files: the state_dict (keys, shapes AND values) of one seeded instance,
                    a seeded input and the forward output(s) -- data only, no reference source text
The tests (tests/test_surface.py) rebuild the same module from rlrep_amd, load the state_dict strictly and compare outputs.
"""
import sys
sys.dont_write_bytecode = True
import os, json, inspect
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch

from make_fixtures import _import_reference, ActionSpace

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, 'surface')


def sig(fn):
    out = []
    for n, p in inspect.signature(fn).parameters.items():
        d = None if p.default is inspect._empty else repr(p.default)
        out.append([n, str(p.kind), d])
    return out


def class_surface(cls, methods):
    s = {'__init__': sig(cls.__init__)}
    for m in methods:
        if hasattr(cls, m):
            f = getattr(cls, m)
            s[m] = 'property' if isinstance(inspect.getattr_static(cls, m), property) else sig(f)
    return s


AGENT_METHODS = ['select_action', 'train', 'critic_step', 'update_actor_and_alpha', 'update_target', 'feature_step',
                 'critic_feeder_feature_step', 'update_feature_target', 'alpha', 'generate_alphabars_and_alphas']


def main():
    mods = _import_reference()
    import importlib
    rutil = importlib.import_module('utils.util')
    rvae = importlib.import_module('networks.vae')
    rcritic = importlib.import_module('networks.critic')
    rpolicy = importlib.import_module('networks.policy')
    ractor = importlib.import_module('agent.sac.actor')
    rsaccritic = importlib.import_module('agent.sac.critic')
    os.makedirs(OUT, exist_ok=True)

    S = {}
    S['agent.sac.sac_agent.SACAgent'] = class_surface(mods['sac'].SACAgent, AGENT_METHODS)
    S['agent.vlsac.vlsac_agent.VLSACAgent'] = class_surface(mods['vlsac'].VLSACAgent, AGENT_METHODS)
    S['agent.ctrlsac.ctrlsac_agent.CTRLSACAgent'] = class_surface(mods['ctrlsac'].CTRLSACAgent, AGENT_METHODS)
    S['agent.spedersac.spedersac_agent.SPEDERSACAgent'] = class_surface(mods['spedersac'].SPEDERSACAgent, AGENT_METHODS)
    S['agent.diffsrsac.diffsrsac_agent.DIFFSRSACAgent'] = class_surface(mods['diffsrsac'].DIFFSRSACAgent, AGENT_METHODS)
    S['utils.buffer.ReplayBuffer'] = class_surface(mods['buffer'].ReplayBuffer, ['add', 'sample'])
    S['utils.buffer.Batch'] = {'_fields': list(mods['buffer'].Batch._fields)}
    for fn in ('unpack_batch', 'mlp', 'weight_init', 'to_np'):
        if hasattr(rutil, fn):
            S['utils.util.' + fn] = sig(getattr(rutil, fn))
    S['utils.util.MLP'] = class_surface(rutil.MLP, ['forward'])
    for m, names in ((rvae, ['Encoder', 'Decoder', 'GaussianFeature']), (rcritic, ['ValueCritic', 'Critic', 'LinearCritic', 'RFFLinearCritic']),
                     (rpolicy, ['GaussianPolicy']), (ractor, ['DiagGaussianActor', 'SquashedNormal', 'TanhTransform']),
                     (rsaccritic, ['DoubleQCritic'])):
        for n in names:
            S[f'{m.__name__}.{n}'] = class_surface(getattr(m, n), ['forward', 'sample', 'rsample', 'log_prob', 'mean'])
    with open(os.path.join(OUT, 'signatures.json'), 'w') as f:
        json.dump(S, f, indent=1, sort_keys=True)

    # ---- module known-answers ------------------------------------------------------------------------------------------------
    torch.set_num_threads(1)
    out, meta = {}, {}
    sd_dim, ad_dim, B = 7, 3, 5
    space = ActionSpace(ad_dim, 2.0)
    space.low[0] = -1.0                                   # asymmetric bounds exercise action_bias

    def record(tag, module, inputs, call='forward', ctor=None):
        for k, v in module.state_dict().items():
            out[f'{tag}/sd/{k}'] = v.detach().numpy()
        meta[tag] = {'ctor': ctor, 'call': call, 'n_in': len(inputs), 'sd_keys': list(module.state_dict().keys())}
        for i, x in enumerate(inputs):
            out[f'{tag}/in/{i}'] = x.numpy()
        with torch.no_grad():
            res = getattr(module, call)(*inputs)
        res = res if isinstance(res, (tuple, list)) else (res,)
        for i, r in enumerate(res):
            out[f'{tag}/out/{i}'] = r.detach().numpy()
        meta[tag]['n_out'] = len(res)

    g = torch.Generator().manual_seed(11)
    rnd = lambda *sh: torch.randn(*sh, generator=g)
    torch.manual_seed(3)
    record('networks.critic.ValueCritic', rcritic.ValueCritic(sd_dim, 12), [rnd(B, sd_dim)], ctor=dict(state_dim=sd_dim, hidden_dim=12))
    record('networks.critic.Critic', rcritic.Critic(sd_dim, ad_dim, 12), [rnd(B, sd_dim), rnd(B, ad_dim)], ctor=dict(state_dim=sd_dim, action_dim=ad_dim, hidden_dim=12))
    record('networks.critic.LinearCritic', rcritic.LinearCritic(9, 12), [rnd(B, 9)], ctor=dict(feature_dim=9, hidden_dim=12))
    rff = rcritic.RFFLinearCritic(9, 20, 12)
    record('networks.critic.RFFLinearCritic', rff, [rnd(B, 9)], ctor=dict(feature_dim=9, num_rff=20, hidden_dim=12))
    meta['networks.critic.RFFLinearCritic']['l1_equals_l4'] = bool(torch.equal(rff.l1.weight, rff.l4.weight) and torch.equal(rff.l1.bias, rff.l4.bias))
    meta['networks.critic.RFFLinearCritic']['bias_range'] = [float(rff.l1.bias.min()), float(rff.l1.bias.max())]
    record('networks.policy.GaussianPolicy', rpolicy.GaussianPolicy(sd_dim, ad_dim, space, 12), [rnd(B, sd_dim)],
           ctor=dict(state_dim=sd_dim, action_dim=ad_dim, hidden_dim=12))
    record('networks.vae.Encoder', rvae.Encoder(sd_dim, ad_dim, 6, 12), [rnd(B, sd_dim), rnd(B, ad_dim), rnd(B, sd_dim)],
           ctor=dict(state_dim=sd_dim, action_dim=ad_dim, feature_dim=6, hidden_dim=12))
    record('networks.vae.Decoder', rvae.Decoder(sd_dim, 6, 12), [rnd(B, 6)], ctor=dict(state_dim=sd_dim, feature_dim=6, hidden_dim=12))
    record('networks.vae.GaussianFeature', rvae.GaussianFeature(sd_dim, ad_dim, 6, 12), [rnd(B, sd_dim), rnd(B, ad_dim)],
           ctor=dict(state_dim=sd_dim, action_dim=ad_dim, feature_dim=6, hidden_dim=12))
    record('agent.sac.critic.DoubleQCritic', rsaccritic.DoubleQCritic(sd_dim, ad_dim, 12, 2), [rnd(B, sd_dim), rnd(B, ad_dim)],
           ctor=dict(obs_dim=sd_dim, action_dim=ad_dim, hidden_dim=12, hidden_depth=2))
    record('utils.util.MLP', rutil.MLP(sd_dim, 12, 4, 2), [rnd(B, sd_dim)], ctor=dict(input_dim=sd_dim, hidden_dim=12, output_dim=4, hidden_depth=2))
    # GaussianPolicy.sample with the noise pinned: x_t = mean + std * eps through Normal.rsample's _standard_normal
    import torch.distributions.normal as tdn
    pol = rpolicy.GaussianPolicy(sd_dim, ad_dim, space, 12)
    x = rnd(B, sd_dim)
    eps = rnd(B, ad_dim)
    orig = tdn._standard_normal
    tdn._standard_normal = lambda shape, dtype, device: eps.clone()
    try:
        with torch.no_grad():
            a, lp, mean = pol.sample(x)
    finally:
        tdn._standard_normal = orig
    for k, v in pol.state_dict().items():
        out[f'policy_sample/sd/{k}'] = v.numpy()
    out['policy_sample/in/0'], out['policy_sample/eps'] = x.numpy(), eps.numpy()
    out['policy_sample/out/0'], out['policy_sample/out/1'], out['policy_sample/out/2'] = a.numpy(), lp.numpy(), mean.numpy()
    meta['policy_sample'] = {'ctor': dict(state_dim=sd_dim, action_dim=ad_dim, hidden_dim=12)}
    # the live actor: distribution object -> (mean, rsample with pinned eps, log_prob)
    act = ractor.DiagGaussianActor(sd_dim, ad_dim, 12, 2, [-5., 2.])
    obs = rnd(B, sd_dim)
    eps2 = rnd(B, ad_dim)
    tdn._standard_normal = lambda shape, dtype, device: eps2.clone()
    try:
        with torch.no_grad():
            d = act(obs)
            y = d.rsample()
            lp = d.log_prob(y).sum(-1, keepdim=True)
            mu = d.mean
    finally:
        tdn._standard_normal = orig
    for k, v in act.state_dict().items():
        out[f'actor/sd/{k}'] = v.numpy()
    out['actor/in/0'], out['actor/eps'] = obs.numpy(), eps2.numpy()
    out['actor/out/0'], out['actor/out/1'], out['actor/out/2'] = mu.numpy(), y.numpy(), lp.numpy()
    meta['actor'] = {'ctor': dict(obs_dim=sd_dim, action_dim=ad_dim, hidden_dim=12, hidden_depth=2, log_std_bounds=[-5., 2.]),
                     'sd_keys': list(act.state_dict().keys())}
    # diffsrsac noise schedule (scipy path of the reference) -- e2
    ag = mods['diffsrsac'].DIFFSRSACAgent(state_dim=5, action_dim=3, action_space=ActionSpace(3, 1.0), feature_dim=8, phi_hidden_dim=16,
                                          nabla_mu_hidden_dim=16, hidden_dim=16)
    out['alphabars/default'] = ag.noise_alphabars.numpy()
    if hasattr(ag, 'noise_alphas'):
        out['alphas/default'] = torch.as_tensor(ag.noise_alphas).numpy()
    # ReplayBuffer ring semantics (utils/buffer.py:28-36): ptr/size after wrap-around, surviving rows
    rb = mods['buffer'].ReplayBuffer(2, 1, max_size=5)
    for i in range(8):
        rb.add(np.full(2, i, np.float32), np.full(1, 10 + i, np.float32), np.full(2, 100 + i, np.float32), float(i), float(i % 2))
    out['ring/state'], out['ring/action'], out['ring/next_state'] = rb.state.copy(), rb.action.copy(), rb.next_state.copy()
    out['ring/reward'], out['ring/done'] = rb.reward.copy(), rb.done.copy()
    meta['ring'] = {'ptr': int(rb.ptr), 'size': int(rb.size), 'max_size': int(rb.max_size), 'adds': 8}
    out['meta/json'] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, 'modules.npz'), **out)
    print('surface:', len(S), 'signature entries;', len(out), 'arrays')


if __name__ == '__main__':
    main()
