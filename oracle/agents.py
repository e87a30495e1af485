"""CPU restatement of the five state-based agents' update path (oracle; test infrastructure only).

Each oracle holds ONE flat dict `P` of tensors keyed exactly like the reference's modules'
`state_dict()` entries prefixed by the agent attribute name (`actor.trunk.0.weight`,
`encoder.mean_linear.bias`, `critic_target.l3.weight`, ...), plus `log_alpha` (float64 0-dim, quirk Q1)
and, for vlsac, `critic.noise` / `critic_target.noise` (quirk Q3).  Noise and sample indices are
explicit arguments (SURVEY.md Appendix B gives the reference's draw order).

Every function cites the reference lines it follows (paths relative to the reference repo root).
Mathematically redundant reference work is deduplicated where the value is unchanged (Q4, Q6, Q10,
Q12): those places are marked `[dedupe]`.
"""
import math
import collections
import torch
import torch.nn.functional as F

from .optim import Adam, polyak

# field order of utils/buffer.py:7-10
Batch = collections.namedtuple('Batch', ['state', 'action', 'reward', 'next_state', 'done'])

LOG_SIG_MAX, LOG_SIG_MIN = 2, -20          # networks/vae.py:9-10
LOG_STD_BOUNDS = (-5., 2.)                 # agent/sac/sac_agent.py:64


# ----------------------------------------------------------------------------------------------
# functional network pieces
# ----------------------------------------------------------------------------------------------
def _lin(P, name, x):
    return F.linear(x, P[name + '.weight'], P[name + '.bias'])


def mlp_elu(P, prefix, x, depth):
    """utils/util.py:85-96  mlp(): Linear(+ELU) * depth, Linear.  Sequential indices 0,2,4,..."""
    for i in range(depth):
        x = F.elu(_lin(P, f'{prefix}.{2 * i}', x))
    return _lin(P, f'{prefix}.{2 * depth}', x)


def actor_mu_std(P, obs, prefix='actor'):
    """agent/sac/actor.py:76-91 DiagGaussianActor.forward (hidden_depth=2)."""
    mu, log_std = mlp_elu(P, prefix + '.trunk', obs, 2).chunk(2, dim=-1)
    log_std = torch.tanh(log_std)
    lo, hi = LOG_STD_BOUNDS
    log_std = lo + 0.5 * (hi - lo) * (log_std + 1)
    return mu, log_std.exp()


def squashed_rsample_logp(mu, std, eps):
    """agent/sac/actor.py:16-60 + torch.distributions Normal/TransformedDistribution (un-vendored torch):
    x = mu + eps*std; y = tanh x; log_prob(y) with the cached pre-tanh x (cache_size=1)."""
    x = mu + eps * std
    y = torch.tanh(x)
    var = std ** 2
    base = -((x - mu) ** 2) / (2 * var) - std.log() - math.log(math.sqrt(2 * math.pi))
    ladj = 2. * (math.log(2.) - x - F.softplus(-2. * x))          # actor.py:40-43
    return y, (base - ladj).sum(-1, keepdim=True)


def double_q(P, prefix, obs, act):
    """agent/sac/critic.py:15-36 DoubleQCritic (hidden_depth=2)."""
    xa = torch.cat([obs, act], dim=-1)
    return mlp_elu(P, prefix + '.Q1', xa, 2), mlp_elu(P, prefix + '.Q2', xa, 2)


def gauss_head(P, prefix, x):
    """networks/vae.py:37-48 / 111-120: l1,l2 ReLU then mean / clamped log_std heads."""
    z = F.relu(_lin(P, prefix + '.l1', x))
    z = F.relu(_lin(P, prefix + '.l2', z))
    mean = _lin(P, prefix + '.mean_linear', z)
    log_std = torch.clamp(_lin(P, prefix + '.log_std_linear', z), min=LOG_SIG_MIN, max=LOG_SIG_MAX)
    return mean, log_std


def vl_decoder(P, z):
    """networks/vae.py:79-86."""
    x = F.relu(_lin(P, 'decoder.l1', z))
    return _lin(P, 'decoder.state_linear', x), _lin(P, 'decoder.reward_linear', x)


def vl_critic(P, prefix, mean, log_std):
    """agent/vlsac/vlsac_agent.py:44-63: noise-averaged critic; BOTH heads end in l3 (quirk Q2)."""
    noise = P[prefix + '.noise']
    std = log_std.exp()
    B, d = mean.shape
    x = (mean[:, None, :] + std[:, None, :] * noise).reshape(-1, d)
    n = noise.shape[0]
    q1 = F.elu(_lin(P, prefix + '.l1', x)).reshape(B, n, -1).mean(dim=1)
    q1 = _lin(P, prefix + '.l3', F.elu(_lin(P, prefix + '.l2', q1)))
    q2 = F.elu(_lin(P, prefix + '.l4', x)).reshape(B, n, -1).mean(dim=1)
    q2 = _lin(P, prefix + '.l3', F.elu(_lin(P, prefix + '.l5', q2)))
    return q1, q2


def ctrl_phi(P, prefix, s, a):
    """agent/ctrlsac/ctrlsac_agent.py:72-77."""
    z = F.elu(_lin(P, prefix + '.l1', torch.cat([s, a], -1)))
    z = F.elu(_lin(P, prefix + '.l2', z))
    return _lin(P, prefix + '.l3', z)


def ctrl_mu(P, s):
    """agent/ctrlsac/ctrlsac_agent.py:96-102 (tanh-bounded output)."""
    z = F.elu(_lin(P, 'mu.l1', s))
    z = F.elu(_lin(P, 'mu.l2', z))
    return torch.tanh(_lin(P, 'mu.l3', z))


def ctrl_critic(P, prefix, z):
    """agent/ctrlsac/ctrlsac_agent.py:41-52."""
    return (_lin(P, prefix + '.l2', F.elu(_lin(P, prefix + '.l1', z))),
            _lin(P, prefix + '.l5', F.elu(_lin(P, prefix + '.l4', z))))


def rff_critic(P, prefix, z):
    """agent/spedersac/spedersac_agent.py:38-50 and agent/diffsrsac/diffsrsac_agent.py:77-85
    (the diffsrsac regulariser is multiplied by lambda=0 and dropped [dedupe Q12])."""
    q1 = _lin(P, prefix + '.l3', F.elu(_lin(P, prefix + '.l2', torch.sin(_lin(P, prefix + '.l1', z)))))
    q2 = _lin(P, prefix + '.l6', F.elu(_lin(P, prefix + '.l5', torch.sin(_lin(P, prefix + '.l4', z)))))
    return q1, q2


# ----------------------------------------------------------------------------------------------
# agents
# ----------------------------------------------------------------------------------------------
class OracleSAC:
    """agent/sac/sac_agent.py:15-188."""
    alg = 'sac'
    actor_lr_div = 1.0

    def __init__(self, S, A, params, lr=3e-4, discount=0.99, target_update_period=2, tau=0.005,
                 auto_entropy_tuning=True, dtype=torch.float32, **extra):
        self.S, self.A = S, A
        self.dtype = dtype
        self.P = {}
        for k, v in params.items():
            t = torch.as_tensor(v).detach().clone()
            if k == 'log_alpha':
                t = t.to(torch.float64)
            elif t.is_floating_point():
                t = t.to(dtype)
            self.P[k] = t
        self.discount, self.tau, self.period = discount, tau, target_update_period
        self.learn_alpha = auto_entropy_tuning
        self.target_entropy = -A
        self.steps = 0
        self.lr = lr
        self.hp = extra
        self.last_grads = {}
        self._make_optimizers()
        for n in self.trainable:
            self.P[n].requires_grad_(True)
        self.P['log_alpha'].requires_grad_(True)

    # -- helpers ---------------------------------------------------------------------------
    def names(self, prefix):
        return [k for k in self.P if k.startswith(prefix + '.') and not k.endswith('noise')]

    def _make_optimizers(self):
        self.opt_critic = Adam(self.names('critic'), self.lr)
        self.opt_actor = Adam(self.names('actor'), self.lr / self.actor_lr_div)
        self.opt_alpha = Adam(['log_alpha'], self.lr / self.actor_lr_div)
        self.trainable = self.names('critic') + self.names('actor')

    @property
    def alpha(self):
        return self.P['log_alpha'].exp()

    def _apply(self, opt, loss, tag):
        ps = [self.P[n] for n in opt.names]
        gs = torch.autograd.grad(loss, ps, allow_unused=True)
        grads = {n: g for n, g in zip(opt.names, gs)}
        self.last_grads[tag] = {n: g.detach().clone() for n, g in grads.items() if g is not None}
        opt.step(self.P, grads)

    def state(self):
        return {k: v.detach().clone() for k, v in self.P.items()}

    # -- feature hooks (overridden) --------------------------------------------------------
    def q_pair(self, prefix, obs, act):
        return double_q(self.P, prefix, obs, act)

    # -- steps -----------------------------------------------------------------------------
    def critic_step(self, batch, eps_next):
        """sac_agent.py:105-135.  info['q2'] repeats q1 (quirk Q13)."""
        P = self.P
        with torch.no_grad():
            mu, std = actor_mu_std(P, batch.next_state)
            a2, logp = squashed_rsample_logp(mu, std, eps_next)
            tq1, tq2 = self.q_pair('critic_target', batch.next_state, a2)
            target_v = torch.min(tq1, tq2) - self.alpha.detach() * logp
            target_q = batch.reward + (1. - batch.done) * self.discount * target_v
        q1, q2 = self.q_pair('critic', batch.state, batch.action)
        loss = F.mse_loss(q1, target_q) + F.mse_loss(q2, target_q)
        self._apply(self.opt_critic, loss, 'critic')
        return {'q_loss': loss.item(), 'q1': q1.mean().item(), 'q2': q1.mean().item()}

    def _actor_q(self, obs, action):
        return self.q_pair('critic', obs, action)

    def update_actor_and_alpha(self, batch, eps_pi):
        """sac_agent.py:138-166 (and the per-agent copies: vlsac :165-198, ctrlsac :295-325,
        spedersac :259-289, diffsrsac :241-269)."""
        mu, std = actor_mu_std(self.P, batch.state)
        action, logp = squashed_rsample_logp(mu, std, eps_pi)
        q1, q2 = self._actor_q(batch.state, action)
        actor_loss = (self.alpha.detach() * logp - torch.min(q1, q2)).mean()
        self._apply(self.opt_actor, actor_loss, 'actor')
        info = {'actor_loss': actor_loss.item()}
        if self.learn_alpha:
            alpha_loss = (self.alpha * (-logp - self.target_entropy).detach()).mean()
            self._apply(self.opt_alpha, alpha_loss, 'alpha')
            info['alpha_loss'] = alpha_loss.item()
            info['alpha'] = self.alpha.item()
        return info

    def update_target(self):
        """sac_agent.py:99-102."""
        if self.steps % self.period == 0:
            polyak(self.P, 'critic', 'critic_target', self.tau)

    def train(self, batches, noise):
        """sac_agent.py:169-188.  batches: [Batch]; noise: [eps_next(B,A), eps_pi(B,A)]."""
        self.steps += 1
        b = batches[0]
        info = self.critic_step(b, noise[0])
        info.update(self.update_actor_and_alpha(b, noise[1]))
        self.update_target()
        return info

    # number of minibatches / noise tensors one train() consumes
    def n_batches(self):
        return 1


class OracleVLSAC(OracleSAC):
    """agent/vlsac/vlsac_agent.py:67-273."""
    alg = 'vlsac'

    def __init__(self, S, A, params, lr=1e-4, feature_tau=0.001, extra_feature_steps=3, use_feature_target=True, **kw):
        self.feature_tau, self.extra = feature_tau, extra_feature_steps
        # vlsac_agent.py:113-114, 176-179, 214-219, 257-258: without a feature target the critic / actor steps read the LIVE f
        # and there is no f_target (no Polyak either)
        self.use_feature_target = bool(use_feature_target)
        self.fnet = 'f_target' if self.use_feature_target else 'f'
        super().__init__(S, A, params, lr=lr, **kw)

    def _make_optimizers(self):
        feat = self.names('encoder') + self.names('decoder') + self.names('f')
        self.opt_feature = Adam(feat, self.lr)
        self.opt_critic = Adam(self.names('critic'), self.lr)
        self.opt_actor = Adam(self.names('actor'), self.lr)
        self.opt_alpha = Adam(['log_alpha'], self.lr)
        self.trainable = feat + self.names('critic') + self.names('actor')

    def feature_step(self, batch, eps_z):
        """vlsac_agent.py:126-162.  [dedupe Q4] the reference runs the encoder twice on the same
        input; one forward feeds both the reparameterised sample and the KL term (same value, and the
        same gradient: the two paths are summed at (mean1, log_std1))."""
        P = self.P
        x = torch.cat([batch.state, batch.action, batch.next_state], -1)
        mean1, log_std1 = gauss_head(P, 'encoder', x)
        z = mean1 + eps_z * log_std1.exp()                                  # networks/vae.py:50-57
        xs, r = vl_decoder(P, z)
        s_loss = 0.5 * F.mse_loss(xs, batch.next_state)
        r_loss = 0.5 * F.mse_loss(r, batch.reward)
        ml_loss = r_loss + s_loss
        mean2, log_std2 = gauss_head(P, 'f', torch.cat([batch.state, batch.action], -1))
        var1, var2 = (2 * log_std1).exp(), (2 * log_std2).exp()
        kl = log_std2 - log_std1 + 0.5 * (var1 + (mean1 - mean2) ** 2) / var2 - 0.5
        loss = (ml_loss + kl).mean()                                         # quirk Q5
        self._apply(self.opt_feature, loss, 'feature')
        return {'vae_loss': loss.item(), 'ml_loss': ml_loss.item(), 'kl_loss': kl.mean().item(),
                's_loss': s_loss.item(), 'r_loss': r_loss.item()}

    def update_feature_target(self):
        """vlsac_agent.py:240-242 (called only with a feature target, :257-258)."""
        if self.use_feature_target:
            polyak(self.P, 'f', 'f_target', self.feature_tau)

    def critic_step(self, batch, eps_next):
        """vlsac_agent.py:201-237."""
        P = self.P
        with torch.no_grad():
            mu, std = actor_mu_std(P, batch.next_state)
            a2, logp = squashed_rsample_logp(mu, std, eps_next)
            mean, log_std = gauss_head(P, self.fnet, torch.cat([batch.state, batch.action], -1))
            nmean, nlog_std = gauss_head(P, self.fnet, torch.cat([batch.next_state, a2], -1))
            nq1, nq2 = vl_critic(P, 'critic_target', nmean, nlog_std)
            next_q = torch.min(nq1, nq2) - self.alpha * logp
            target_q = batch.reward + (1. - batch.done) * self.discount * next_q
        q1, q2 = vl_critic(P, 'critic', mean, log_std)
        q1_loss, q2_loss = F.mse_loss(target_q, q1), F.mse_loss(target_q, q2)
        self._apply(self.opt_critic, q1_loss + q2_loss, 'critic')
        return {'q1_loss': q1_loss.item(), 'q2_loss': q2_loss.item(),
                'q1': q1.mean().item(), 'q2': q2.mean().item()}

    def _actor_q(self, obs, action):
        mean, log_std = gauss_head(self.P, self.fnet, torch.cat([obs, action], -1))
        return vl_critic(self.P, 'critic', mean, log_std)

    def train(self, batches, noise):
        """vlsac_agent.py:245-273.  batches: extra+1 Batches; noise: [eps_z]*(extra+1), eps_next, eps_pi."""
        self.steps += 1
        n = self.extra + 1
        for i in range(n):
            info = self.feature_step(batches[i], noise[i])
            self.update_feature_target()
        b = batches[n - 1]
        info.update(self.critic_step(b, noise[n]))
        info.update(self.update_actor_and_alpha(b, noise[n + 1]))
        self.update_target()
        return info

    def n_batches(self):
        return self.extra + 1


class OracleCTRLSAC(OracleSAC):
    """agent/ctrlsac/ctrlsac_agent.py:123-361."""
    alg = 'ctrlsac'
    actor_lr_div = 3.0                                   # ctrlsac_agent.py:195-197

    def __init__(self, S, A, params, lr=1e-4, feature_tau=0.005, extra_feature_steps=3, use_feature_target=True, **kw):
        self.feature_tau, self.extra = feature_tau, extra_feature_steps
        self.use_feature_target = bool(use_feature_target)      # ctrlsac_agent.py:167-168, 185-186, 268-273, 340-346
        super().__init__(S, A, params, lr=lr, **kw)

    def _make_optimizers(self):
        feat = self.names('phi') + self.names('mu') + self.names('theta')
        self.opt_feature = Adam(feat, self.lr)
        self.opt_critic = Adam(self.names('critic'), self.lr)
        self.opt_actor = Adam(self.names('actor'), self.lr / 3)
        self.opt_alpha = Adam(['log_alpha'], self.lr / 3)
        self.trainable = feat + self.names('critic') + self.names('actor')

    def feature_step(self, batch):
        """ctrlsac_agent.py:213-251.  [dedupe Q6] the score matrix is phi @ mu'^T instead of the
        [B,B,F] broadcast-multiply-sum; CrossEntropyLoss with identity probability targets (Q7) is
        mean_i(logsumexp_j S_ij - S_ii)."""
        P = self.P
        z_phi = ctrl_phi(P, 'phi', batch.state, batch.action)
        z_mu = ctrl_mu(P, batch.next_state)
        S = z_phi @ z_mu.T
        model_loss = (torch.logsumexp(S, dim=1) - torch.diagonal(S)).mean()
        r_loss = 0.5 * F.mse_loss(_lin(P, 'theta.l', z_phi), batch.reward)
        loss = model_loss + r_loss
        self._apply(self.opt_feature, loss, 'feature')
        return {'total_loss': loss.item(), 'model_loss': model_loss.item(), 'r_loss': r_loss.item()}

    def update_feature_target(self):
        """ctrlsac_agent.py:253-255 (called only with a feature target, :340-341)."""
        if self.use_feature_target:
            polyak(self.P, 'phi', 'phi_target', self.feature_tau)

    def sync_frozen(self):
        """ctrlsac_agent.py:344-346: BOTH frozen copies are loaded from `phi` (quirk Q8)."""
        with torch.no_grad():
            for n in self.names('phi'):
                for dst in ('frozen_phi', 'frozen_phi_target'):
                    k = dst + n[len('phi'):]
                    if k in self.P:
                        self.P[k].copy_(self.P[n])

    def critic_step(self, batch, eps_next):
        """ctrlsac_agent.py:257-293."""
        P = self.P
        with torch.no_grad():
            mu, std = actor_mu_std(P, batch.next_state)
            a2, logp = squashed_rsample_logp(mu, std, eps_next)
            fz = 'frozen_phi_target' if self.use_feature_target else 'frozen_phi'
            z = ctrl_phi(P, fz, batch.state, batch.action)
            z2 = ctrl_phi(P, fz, batch.next_state, a2)
            nq1, nq2 = ctrl_critic(P, 'critic_target', z2)
            target_q = batch.reward + (1. - batch.done) * self.discount * (torch.min(nq1, nq2) - self.alpha * logp)
        q1, q2 = ctrl_critic(P, 'critic', z)
        q1_loss, q2_loss = F.mse_loss(target_q, q1), F.mse_loss(target_q, q2)
        self._apply(self.opt_critic, q1_loss + q2_loss, 'critic')
        return {'q1_loss': q1_loss.item(), 'q2_loss': q2_loss.item(),
                'q1': q1.mean().item(), 'q2': q2.mean().item()}

    def _actor_q(self, obs, action):
        return ctrl_critic(self.P, 'critic', ctrl_phi(self.P, 'frozen_phi', obs, action))

    def train(self, batches, noise):
        """ctrlsac_agent.py:327-361.  noise: [eps_next, eps_pi]."""
        self.steps += 1
        n = self.extra + 1
        for i in range(n):
            info = self.feature_step(batches[i])
            self.update_feature_target()
        self.sync_frozen()
        b = batches[n - 1]
        info.update(self.critic_step(b, noise[0]))
        info.update(self.update_actor_and_alpha(b, noise[1]))
        self.update_target()
        return info

    def n_batches(self):
        return self.extra + 1


class OracleSPEDERSAC(OracleSAC):
    """agent/spedersac/spedersac_agent.py:97-322."""
    alg = 'spedersac'

    def __init__(self, S, A, params, phi_and_mu_lr=1e-5, critic_and_actor_lr=3e-4, feature_tau=0.005,
                 extra_feature_steps=5, phi_hidden_depth=1, mu_hidden_depth=0, use_feature_target=True, **kw):
        self.feature_tau, self.extra = feature_tau, extra_feature_steps
        self.use_feature_target = bool(use_feature_target)      # spedersac_agent.py:150-151, 306-307 (phi_target is never read)
        self.feat_lr = phi_and_mu_lr
        self.phi_depth, self.mu_depth = phi_hidden_depth, mu_hidden_depth
        kw.pop('lr', None)
        super().__init__(S, A, params, lr=critic_and_actor_lr, **kw)

    def _make_optimizers(self):
        feat = self.names('phi') + self.names('mu') + self.names('theta')
        self.opt_feature = Adam(feat, self.feat_lr)
        self.opt_critic = Adam(self.names('critic'), self.lr)
        self.opt_actor = Adam(self.names('actor'), self.lr)
        self.opt_alpha = Adam(['log_alpha'], self.lr)
        self.trainable = feat + self.names('critic') + self.names('actor')

    def phi(self, s, a):
        return mlp_elu(self.P, 'phi.trunk', torch.cat([s, a], -1), self.phi_depth)

    def mu(self, s):
        return mlp_elu(self.P, 'mu.trunk', s, self.mu_depth)

    def feature_step(self, batch, batch_r):
        """spedersac_agent.py:181-219.  [dedupe Q10]
        mean(-2 diag(phi mu'^T)) = -(2/B) sum_i phi_i.mu'_i ;
        mean((phi_r mu_r^T)(phi_r mu_r^T)^T) = (1/B^2) sum_k (Phi_bar . mu_r,k)^2, Phi_bar = sum_i phi_r,i."""
        P = self.P
        B = batch.state.shape[0]
        z_phi = self.phi(batch.state, batch.action)
        z_phi_r = self.phi(batch_r.state, batch_r.action)
        z_mu = self.mu(batch.next_state)
        z_mu_r = self.mu(batch_r.next_state)
        pt1 = -2.0 * (z_phi * z_mu).sum(-1).sum() / B
        c = z_mu_r @ z_phi_r.sum(0)
        pt2 = (c * c).sum() / (B * B)
        model_loss = pt1 + pt2
        r_loss = 0.5 * F.mse_loss(_lin(P, 'theta.l', z_phi), batch.reward)
        loss = model_loss + r_loss
        self._apply(self.opt_feature, loss, 'feature')
        return {'total_loss': loss.item(), 'model_loss': model_loss.item(), 'r_loss': r_loss.item()}

    def update_feature_target(self):
        """spedersac_agent.py:221-223 (called only with a feature target, :306-307)."""
        if self.use_feature_target:
            polyak(self.P, 'phi', 'phi_target', self.feature_tau)

    def critic_step(self, batch, eps_next):
        """spedersac_agent.py:225-257 (critic reads the LIVE phi under no_grad)."""
        P = self.P
        with torch.no_grad():
            mu, std = actor_mu_std(P, batch.next_state)
            a2, logp = squashed_rsample_logp(mu, std, eps_next)
            z = self.phi(batch.state, batch.action)
            z2 = self.phi(batch.next_state, a2)
            nq1, nq2 = rff_critic(P, 'critic_target', z2)
            target_q = batch.reward + (1. - batch.done) * self.discount * (torch.min(nq1, nq2) - self.alpha * logp)
        q1, q2 = rff_critic(P, 'critic', z)
        q1_loss, q2_loss = F.mse_loss(target_q, q1), F.mse_loss(target_q, q2)
        self._apply(self.opt_critic, q1_loss + q2_loss, 'critic')
        return {'q1_loss': q1_loss.item(), 'q2_loss': q2_loss.item(),
                'q1': q1.mean().item(), 'q2': q2.mean().item()}

    def _actor_q(self, obs, action):
        return rff_critic(self.P, 'critic', self.phi(obs, action))

    def train(self, batches, noise):
        """spedersac_agent.py:291-322.  batches: 2*(extra+1) in draw order (batch_1, batch_2, ...)."""
        self.steps += 1
        n = self.extra + 1
        for i in range(n):
            info = self.feature_step(batches[2 * i], batches[2 * i + 1])
            self.update_feature_target()
        b = batches[2 * (n - 1)]
        info.update(self.critic_step(b, noise[0]))
        info.update(self.update_actor_and_alpha(b, noise[1]))
        self.update_target()
        return info

    def n_batches(self):
        return 2 * (self.extra + 1)


class OracleDIFFSRSAC(OracleSAC):
    """agent/diffsrsac/diffsrsac_agent.py:93-343."""
    alg = 'diffsrsac'

    def __init__(self, S, A, params, feature_dim=256, phi_and_nabla_mu_lr=0.003, critic_and_actor_lr=3e-4,
                 extra_feature_steps=3, sigma_scale_factor=0.449, phi_hidden_depth=1,
                 nabla_mu_hidden_depth=1, critic_elu_layer_regularizer_lambda=0, **kw):
        self.extra = extra_feature_steps
        self.reg_lambda = float(critic_elu_layer_regularizer_lambda)
        self.feat_lr = phi_and_nabla_mu_lr
        self.sigma = sigma_scale_factor
        self.F = feature_dim
        self.phi_depth, self.nm_depth = phi_hidden_depth, nabla_mu_hidden_depth
        kw.pop('lr', None)
        super().__init__(S, A, params, lr=critic_and_actor_lr, **kw)

    def _make_optimizers(self):
        phi, nm = self.names('critic_feed_feature'), self.names('nablamu_net')
        self.opt_phi = Adam(phi, self.feat_lr)
        self.opt_nm = Adam(nm, self.feat_lr)
        # quirk Q11: the reference's critic_optimizer holds the discarded DoubleQCritic's parameters, so the
        # RFF critic is never updated: no critic optimizer here at all.
        self.opt_actor = Adam(self.names('actor'), self.lr)
        self.opt_alpha = Adam(['log_alpha'], self.lr)
        self.trainable = phi + nm + self.names('actor')

    def phi(self, s, a):
        return mlp_elu(self.P, 'critic_feed_feature.z_vector', torch.cat([s, a], -1), self.phi_depth)

    def critic_feeder_feature_step(self, batch, noise_idx, eps_s):
        """diffsrsac_agent.py:271-318.  noise_idx: int64[B]; eps_s: [B,S] ALREADY scaled by
        sigma_scale_factor (the reference draws torch.normal(0, 0.449))."""
        P = self.P
        B = batch.state.shape[0]
        ab = P['noise_alphabars'].index_select(0, noise_idx).reshape(B, 1)
        pert = torch.sqrt(ab) * batch.next_state + torch.sqrt(1.0 - ab) * eps_s
        tgt = -(pert - torch.sqrt(ab) * batch.next_state)
        phi = self.phi(batch.state, batch.action)
        u = mlp_elu(P, 'nablamu_net.Mu_z_by_s_layer', torch.cat([pert, ab], -1), self.nm_depth)
        score = torch.bmm(phi.unsqueeze(1), u.reshape(B, self.F, self.S)).squeeze(1)
        diff = tgt - (1. - ab) * self.sigma * score
        loss = ((1.0 / B) * (diff ** 2).sum(dim=1)).sum()
        # two optimizers, one backward (diffsrsac_agent.py:308-314)
        names = self.opt_nm.names + self.opt_phi.names
        gs = torch.autograd.grad(loss, [P[n] for n in names])
        grads = dict(zip(names, gs))
        self.last_grads['nablamu'] = {n: grads[n].detach().clone() for n in self.opt_nm.names}
        self.last_grads['phi'] = {n: grads[n].detach().clone() for n in self.opt_phi.names}
        self.opt_nm.step(P, grads)
        self.opt_phi.step(P, grads)
        return {'score_loss': loss.item()}

    def _reg_terms(self, prefix, z):
        """diffsrsac_agent.py:62-90: per head, x = l2(elu(l2(sin(l1 z)))) -- the SECOND linear layer applied once more to its own ELU
        output -- and lambda * ( (sum_{i != j} (x_i . x_j)^2) / ((n - 1) n) - 2 mean_i |x_i|^2 / d + 1 / d )."""
        P, lam = self.P, self.reg_lambda
        total = 0.
        for l1, l2 in (('.l1', '.l2'), ('.l4', '.l5')):
            e = F.elu(_lin(P, prefix + l2, torch.sin(_lin(P, prefix + l1, z))))
            x = _lin(P, prefix + l2, e)
            n, d = x.shape
            inprods = x @ x.T
            norms = torch.diagonal(inprods)
            part1 = (inprods.pow(2).sum() - norms.pow(2).sum()) / ((n - 1) * n)
            total = total + lam * (part1 - 2. * norms.mean() / d + 1. / d)
        return total

    def critic_step(self, batch, eps_next):
        """diffsrsac_agent.py:205-239: the loss is evaluated but no parameter moves (quirk Q11); the regulariser of the target AND the
        live critic enters q_loss_reg only (0 at the default lambda, Q12); info['q2'] repeats q1 (Q13)."""
        P = self.P
        with torch.no_grad():
            mu, std = actor_mu_std(P, batch.next_state)
            a2, logp = squashed_rsample_logp(mu, std, eps_next)
            zn, zc = self.phi(batch.next_state, a2), self.phi(batch.state, batch.action)
            tq1, tq2 = rff_critic(P, 'critic_target', zn)
            target_q = batch.reward + (1. - batch.done) * self.discount * (torch.min(tq1, tq2) - self.alpha.detach() * logp)
            q1, q2 = rff_critic(P, 'critic', zc)
            loss = F.mse_loss(q1, target_q) + F.mse_loss(q2, target_q)
            reg = 0.
            if self.reg_lambda != 0:
                reg = self._reg_terms('critic_target', zn) + self._reg_terms('critic', zc)
        return {'q_loss_reg': float(loss + reg), 'q_loss_noreg': loss.item(), 'q1': q1.mean().item(), 'q2': q1.mean().item()}

    def _actor_q(self, obs, action):
        return rff_critic(self.P, 'critic', self.phi(obs, action))

    def update_target(self):
        """sac_agent.py:99-102 on RFFCritic pairs whose values never diverge (load_state_dict at
        diffsrsac_agent.py:168, critic never trained): still executed for fidelity."""
        if self.steps % self.period == 0:
            polyak(self.P, 'critic', 'critic_target', self.tau)

    def train(self, batches, noise):
        """diffsrsac_agent.py:320-343.  noise: [(idx_i, eps_s_i)]*(extra+1) flattened, eps_next, eps_pi."""
        self.steps += 1
        n = self.extra + 1
        for i in range(n):
            info = self.critic_feeder_feature_step(batches[i], noise[2 * i], noise[2 * i + 1])
        b = batches[n - 1]
        info.update(self.critic_step(b, noise[2 * n]))
        info.update(self.update_actor_and_alpha(b, noise[2 * n + 1]))
        self.update_target()
        return info

    def n_batches(self):
        return self.extra + 1


_CLASSES = {'sac': OracleSAC, 'vlsac': OracleVLSAC, 'ctrlsac': OracleCTRLSAC,
            'spedersac': OracleSPEDERSAC, 'diffsrsac': OracleDIFFSRSAC}


def make_oracle(alg, S, A, params, **hp):
    """hp: the reference constructor's keyword arguments (unknown ones are ignored)."""
    hp = dict(hp)
    if alg not in ('vlsac', 'ctrlsac', 'spedersac'):
        hp.pop('use_feature_target', None)
    if alg != 'diffsrsac':
        hp.pop('critic_elu_layer_regularizer_lambda', None)
    for k in ('hidden_dim', 'alpha', 'phi_hidden_dim', 'mu_hidden_dim',
              'critic_and_actor_hidden_dim', 'nabla_mu_hidden_dim', 'num_noises',
              'DARL_noise_a', 'DARL_noise_b', 'action_space',
              'state_dim', 'action_dim'):
        hp.pop(k, None)
    if alg in ('sac', 'vlsac', 'ctrlsac'):
        hp.pop('feature_dim', None)
    if alg == 'spedersac':
        hp.pop('feature_dim', None)
    return _CLASSES[alg](S, A, params, **hp)


def gather_batch(replay, idx, dtype=torch.float32):
    """utils/buffer.py:39-48: fancy-index gather + float32 cast."""
    idx = torch.as_tensor(idx, dtype=torch.long)
    return Batch(*(torch.as_tensor(replay[k]).to(dtype)[idx]
                   for k in ('state', 'action', 'reward', 'next_state', 'done')))
