"""CTRLSACAgent (reference agent/ctrlsac/ctrlsac_agent.py:123-361) on the HIP step programs.

feature_step = InfoNCE over the BxB score matrix phi(s,a).mu(s')^T (a GEMM, not the reference's [B,B,F]
broadcast: quirk Q6) + 0.5*mse(theta(phi), r), Adam and Polyak phi->phi_target in one step program.
"""
from rlrep_amd import _lib
from rlrep_amd.agent.sac.sac_agent import SACAgent, device  # noqa: F401


class CTRLSACAgent(SACAgent):
    ALG = 'ctrlsac'
    MODULES = ('critic', 'critic_target', 'actor', 'phi', 'phi_target', 'mu', 'theta', 'frozen_phi', 'frozen_phi_target')
    FEATURE_KEYS = ('total_loss', 'model_loss', 'r_loss')
    CRITIC_KEYS = ('q1_loss', 'q2_loss', 'q1', 'q2')

    def __init__(self, state_dim, action_dim, action_space, lr=1e-4, discount=0.99, target_update_period=2,
                 tau=0.005, alpha=0.1, auto_entropy_tuning=True, hidden_dim=1024, feature_tau=0.005,
                 feature_dim=2048, use_feature_target=True, extra_feature_steps=1, **_hip):
        self._init_common(state_dim, action_dim, action_space, discount, target_update_period, tau, alpha,
                          auto_entropy_tuning)
        self.feature_dim, self.feature_tau = int(feature_dim), float(feature_tau)
        self.use_feature_target = bool(use_feature_target)
        if not self.use_feature_target:
            # ctrlsac_agent.py:167-168, 185-186: no phi_target / frozen_phi_target attributes; the critic step reads frozen_phi (:268-273)
            self.MODULES = tuple(m for m in self.MODULES if m not in ('phi_target', 'frozen_phi_target'))
        self.extra_feature_steps = int(extra_feature_steps)
        # actor hidden is hard-coded to 256 (ctrlsac_agent.py:188-194); actor and alpha use lr/3 (:195-197)
        self._dims = dict(state_dim=state_dim, action_dim=action_dim, hidden_dim=hidden_dim, actor_hidden_dim=256,
                          feature_dim=feature_dim, phi_hidden_dim=hidden_dim, phi_hidden_depth=2,
                          mu_hidden_dim=hidden_dim, mu_hidden_depth=2,
                          flags=0 if self.use_feature_target else _lib.FLAG_NO_FEATURE_TARGET)
        self._hyper = dict(lr_feature=lr, lr_critic=lr, lr_actor=lr / 3)
        self._finish_init(_hip)

    def _init_parameters(self):
        self._init_prefix('actor', True)
        for m in ('phi', 'mu', 'theta', 'critic', 'frozen_phi'):
            self._init_prefix(m, False)
        self._copy_prefix('phi', 'phi_target')                       # ctrlsac_agent.py:164-165
        self._copy_prefix('critic', 'critic_target')                 # :201
        self._copy_prefix('frozen_phi', 'frozen_phi_target')         # :185-186

    def feature_step(self, batch):
        """ctrlsac_agent.py:213-251 (+ update_feature_target :253-255 fused into the optimizer launch)."""
        self.flush()          # nothing of a pipelined train() may still be reading the slot / writing the feature parameters
        self._set_batch(batch)
        self.core.feature_step(None)
        return self.core.info(self.FEATURE_KEYS)

    def update_feature_target(self):
        return None

    def _feature_iters(self):
        return self.extra_feature_steps + 1

    def _plan(self, B):
        return ([f'f{i}' for i in range(self._feature_iters())],
                [('crit', (B, self.action_dim)), ('act', (B, self.action_dim))])

    def _feature_once(self, buffer, B, i, g):
        c = self.core
        self._sample_into(buffer, B, f'f{i}', 0, g)
        if self.world_size > 1:
            self._feature_backward_dp(None); self._allreduce(0); c.feature_apply()
        else:
            c.feature_step(None)

    def _between_feature_and_critic(self):
        self.core.sync_frozen()                                      # ctrlsac_agent.py:344-346 (quirk Q8)
