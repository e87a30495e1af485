"""SPEDERSACAgent (reference agent/spedersac/spedersac_agent.py:97-322) on the HIP step programs.

feature_step = spectral-decomposition loss in its O(B*F) form (quirk Q10) on two minibatches + reward head,
Adam and Polyak phi->phi_target; critic = RFF critic sin/ELU on the LIVE phi.
"""
from rlrep_amd import _lib
from rlrep_amd.agent.sac.sac_agent import SACAgent, device  # noqa: F401


class SPEDERSACAgent(SACAgent):
    ALG = 'spedersac'
    PREFETCH_CHAIN = False    # two slots per feature step: batches are gathered by plain replay_sample calls
    MODULES = ('critic', 'critic_target', 'actor', 'phi', 'phi_target', 'mu', 'theta')
    FEATURE_KEYS = ('total_loss', 'model_loss', 'r_loss')
    CRITIC_KEYS = ('q1_loss', 'q2_loss', 'q1', 'q2')

    def __init__(self, state_dim, action_dim, action_space, phi_and_mu_lr=-1, phi_hidden_dim=-1, phi_hidden_depth=-1,
                 mu_hidden_dim=-1, mu_hidden_depth=-1, critic_and_actor_lr=-1, critic_and_actor_hidden_dim=-1,
                 discount=0.99, target_update_period=2, tau=0.005, alpha=0.1, auto_entropy_tuning=True,
                 hidden_dim=1024, feature_tau=0.005, feature_dim=2048, use_feature_target=True,
                 extra_feature_steps=1, **_hip):
        self._init_common(state_dim, action_dim, action_space, discount, target_update_period, tau, alpha,
                          auto_entropy_tuning)
        self.feature_dim, self.feature_tau = int(feature_dim), float(feature_tau)
        self.use_feature_target = bool(use_feature_target)
        if not self.use_feature_target:
            self.MODULES = tuple(m for m in self.MODULES if m != 'phi_target')       # spedersac_agent.py:150-151
        self.extra_feature_steps = int(extra_feature_steps)
        self._dims = dict(state_dim=state_dim, action_dim=action_dim, hidden_dim=critic_and_actor_hidden_dim,
                          actor_hidden_dim=critic_and_actor_hidden_dim, feature_dim=feature_dim,
                          phi_hidden_dim=max(phi_hidden_dim, 1), phi_hidden_depth=phi_hidden_depth,
                          mu_hidden_dim=max(mu_hidden_dim, 1), mu_hidden_depth=mu_hidden_depth,
                          flags=0 if self.use_feature_target else _lib.FLAG_NO_FEATURE_TARGET)
        self._hyper = dict(lr_feature=phi_and_mu_lr, lr_critic=critic_and_actor_lr, lr_actor=critic_and_actor_lr)
        self._finish_init(_hip)

    def _init_parameters(self):
        self._init_prefix('actor', True)
        for m in ('phi', 'mu', 'theta', 'critic'):                   # local MLP: no orthogonal init (quirk Q17)
            self._init_prefix(m, False)
        self._copy_prefix('phi', 'phi_target')
        self._copy_prefix('critic', 'critic_target')

    def feature_step(self, batch, s_random, a_random, s_prime_random):
        """spedersac_agent.py:181-219: the second batch's (s, a, s') are the "random" marginals."""
        import torch
        self.flush()          # nothing of a pipelined train() may still be reading the slot / writing the feature parameters
        self._set_batch(batch, 0)
        z = torch.zeros(s_random.shape[0], 1, device=s_random.device)
        self.core.set_batch(1, s_random, a_random, z, s_prime_random, z)
        self.core.feature_step(None)
        return self.core.info(self.FEATURE_KEYS)

    def update_feature_target(self):
        return None

    def _feature_iters(self):
        return self.extra_feature_steps + 1

    def _plan(self, B):
        keys = []
        for i in range(self._feature_iters()):
            keys += [f'f{i}a', f'f{i}b']
        return keys, [('crit', (B, self.action_dim)), ('act', (B, self.action_dim))]

    def _prefetch_chain(self, idx_keys):
        # two chains, one per slot: f{i}a -> f{i+1}a (slot 0), f{i}b -> f{i+1}b (slot 1): both gathers of the next feature step ride in this
        # step's optimizer launch (rlrep_prefetch_batch_slot); the critic / actor steps reuse the last pair
        a = [k for k in idx_keys if k.endswith('a')]
        b = [k for k in idx_keys if k.endswith('b')]
        return {**dict(zip(a, a[1:])), **dict(zip(b, b[1:]))}

    def _feature_once(self, buffer, B, i, g):
        c = self.core
        self._sample_into(buffer, B, f'f{i}a', 0, g)
        self._sample_into(buffer, B, f'f{i}b', 1, g)
        if self.world_size > 1:
            self._feature_backward_dp(None); self._allreduce(0); c.feature_apply()
        else:
            c.feature_step(None)
