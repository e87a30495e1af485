"""DIFFSRSACAgent (reference agent/diffsrsac/diffsrsac_agent.py:93-343) on the HIP step programs.

critic_feeder_feature_step = denoising score matching with the factored score phi(s,a)^T grad-mu(s~', abar);
the RFF critic is never trained (quirk Q11): critic_step only evaluates its loss.
"""
import numpy as np
import torch

from rlrep_amd.agent.sac.sac_agent import SACAgent, device  # noqa: F401


def _beta_cdf(x, a, b, n=200001):
    """Regularised incomplete beta I_x(a,b) without scipy (the reference uses scipy.stats.beta.cdf at
    diffsrsac_agent.py:197).  Substitution t = u^(1/a) removes the t^(a-1) endpoint singularity; the
    (1-t)^(b-1) one is handled by symmetry I_x(a,b) = 1 - I_{1-x}(b,a) on the upper half."""
    from math import lgamma, exp
    x = np.asarray(x, dtype=np.float64)

    def lower(xx, aa, bb):
        # int_0^xx t^(aa-1) (1-t)^(bb-1) dt, xx <= 0.5, via u = t^aa
        out = np.zeros_like(xx)
        for i, xv in enumerate(xx):
            if xv <= 0:
                continue
            u = np.linspace(0.0, xv ** aa, n)
            f = (1.0 - u ** (1.0 / aa)) ** (bb - 1.0) / aa
            out[i] = np.sum((f[1:] + f[:-1]) * np.diff(u)) * 0.5          # trapezoid rule
        return out
    Bab = exp(lgamma(a) + lgamma(b) - lgamma(a + b))
    res = np.empty_like(x)
    lo = x <= 0.5
    res[lo] = lower(x[lo], a, b) / Bab
    res[~lo] = 1.0 - lower(1.0 - x[~lo], b, a) / Bab
    return res


def _raw_alphabars(a, b, num_alphas):
    x = np.linspace(0, 1, num_alphas)
    try:
        from scipy.stats import beta
        cdf = beta.cdf(x, a, b)
    except Exception:
        cdf = _beta_cdf(x, a, b)
    return 1. - cdf


def generate_alphabars(a, b, num_alphas):
    """diffsrsac_agent.py:178-203, the noise_alphabars half: 1 - BetaCDF(x; a, b) on a uniform grid, clipped to its second and
    second-to-last values (so that neither alphabar = 1 nor alphabar = 0 is ever drawn)."""
    raw = _raw_alphabars(a, b, num_alphas)
    return np.clip(raw, a_min=raw[-2], a_max=raw[1]).astype(np.float32)


def generate_alphas(a, b, num_alphas, max_beta=0.99):
    """The noise_alphas half (never read by training, diffsrsac_agent.py:150-153): per-step alpha_i = 1 - beta_i with
    beta_i = min(1 - abar_{i+1} / abar_i, 0.99) taken on the UNCLIPPED schedule, the first beta repeated in front."""
    raw = _raw_alphabars(a, b, num_alphas)
    betas = np.minimum(1.0 - raw[1:] / raw[:-1], max_beta)
    return (1.0 - np.concatenate([betas[:1], betas])).astype(np.float32)


class DIFFSRSACAgent(SACAgent):
    ALG = 'diffsrsac'
    MODULES = ('critic', 'critic_target', 'actor', 'critic_feed_feature', 'nablamu_net')
    FEATURE_KEYS = ('score_loss',)
    CRITIC_KEYS = ('q_loss_reg', 'q_loss_noreg', 'q1', 'q2')

    def __init__(self, state_dim, action_dim, action_space, feature_dim=256, phi_and_nabla_mu_lr=0.003,
                 phi_hidden_dim=256, phi_hidden_depth=1, nabla_mu_hidden_dim=512, nabla_mu_hidden_depth=1,
                 critic_and_actor_lr=3e-4, discount=0.99, target_update_period=2, tau=0.005, alpha=0.1,
                 auto_entropy_tuning=True, hidden_dim=256, extra_feature_steps=3, num_noises=1000,
                 critic_elu_layer_regularizer_lambda=0, DARL_noise_a=0.3, DARL_noise_b=0.1,
                 sigma_scale_factor=0.449, **_hip):
        self._init_common(state_dim, action_dim, action_space, discount, target_update_period, tau, alpha,
                          auto_entropy_tuning)
        self.feature_dim = int(feature_dim)
        self.extra_feature_steps = int(extra_feature_steps)
        self.num_noises = int(num_noises)
        self.sigma_scale_factor = float(sigma_scale_factor)
        self._ab = (DARL_noise_a, DARL_noise_b)
        self._dims = dict(state_dim=state_dim, action_dim=action_dim, hidden_dim=hidden_dim, actor_hidden_dim=hidden_dim,
                          feature_dim=feature_dim, phi_hidden_dim=phi_hidden_dim, phi_hidden_depth=phi_hidden_depth,
                          mu_hidden_dim=nabla_mu_hidden_dim, mu_hidden_depth=nabla_mu_hidden_depth, num_noise=num_noises)
        self._hyper = dict(lr_feature=phi_and_nabla_mu_lr, lr_critic=critic_and_actor_lr, lr_actor=critic_and_actor_lr,
                           sigma_scale=sigma_scale_factor, critic_reg_lambda=float(critic_elu_layer_regularizer_lambda))
        self._finish_init(_hip)

    def _init_parameters(self):
        self._init_prefix('actor', True)
        for m in ('critic_feed_feature', 'nablamu_net', 'critic'):   # util.mlp without .apply(weight_init)
            self._init_prefix(m, False)
        self._copy_prefix('critic', 'critic_target')                 # diffsrsac_agent.py:168
        ab = generate_alphabars(self._ab[0], self._ab[1], self.num_noises)
        self.core.view('noise_alphabars').reshape(-1).copy_(torch.from_numpy(ab).reshape(-1))

    @staticmethod
    def generate_alphabars_and_alphas(a, b, num_alphas):
        """diffsrsac_agent.py:178-203: -> (alphabars float32[num_alphas], alphas float32[num_alphas]) as CPU tensors."""
        return torch.from_numpy(generate_alphabars(a, b, num_alphas)), torch.from_numpy(generate_alphas(a, b, num_alphas))

    @property
    def noise_alphabars(self):
        return self.core.view('noise_alphabars').reshape(-1)

    def critic_feeder_feature_step(self, batch, noise_idx=None, eps=None):
        """diffsrsac_agent.py:271-318."""
        self.flush()          # nothing of a pipelined train() may still be reading the slot / writing the feature parameters
        self._set_batch(batch)
        B = self._B
        if noise_idx is None:
            noise_idx = self._indices('nidx', B, self.num_noises)
        if eps is None:
            eps = self._noise('feat', (B, self.state_dim), self.sigma_scale_factor)
        self.core.feature_step(eps, noise_idx)
        return self.core.info(self.FEATURE_KEYS)

    def _critic_trains(self):
        return False

    def _feature_iters(self):
        return self.extra_feature_steps + 1

    def _plan(self, B):
        # the noise-level indices and the sigma-scaled perturbations are drawn per step (not pooled)
        return ([f'f{i}' for i in range(self._feature_iters())],
                [('crit', (B, self.action_dim)), ('act', (B, self.action_dim))])

    def _feature_once(self, buffer, B, i, g):
        c = self.core
        self._sample_into(buffer, B, f'f{i}', 0, g)
        if self._inject is not None:
            nidx = self._eps(f'nidx{i}', (B,), g)
        elif g:
            nidx = self._buf(f'idx_n{i}', (B,), torch.int32)
            c.fill_indices_dev(nidx, self._num_noises_dev(), self._seed, self._graph_off(f'n{i}'))
        else:
            nidx = self._indices(f'n{i}', B, self.num_noises)
        eps = self._eps(f'pert{i}', (B, self.state_dim), g, std=self.sigma_scale_factor)
        if self._dp:
            # the nabla-mu head's gradient slice (99 % of group 3) is all-reduced from INSIDE the backward, as soon as its dW launch is done
            # (exchange kind 3, csrc/agents2.hip build_diffsrsac); the rest of group 3 and group 0 follow the backward
            self._feature_backward_dp(eps, nidx); self._allreduce_rest(3); self._allreduce(0); c.feature_apply()
        else:
            c.feature_step(eps, nidx)

    def _num_noises_dev(self):
        t = self._bufs.get('num_noises')
        if t is None:
            t = self._bufs['num_noises'] = torch.full((1,), self.num_noises, dtype=torch.int32, device=self.core.device)
        return t
