"""Module signatures of the reference's networks/vae.py:13-120 (Encoder, Decoder, GaussianFeature).

Inside the agents these networks live in the flat parameter arenas and are evaluated by the HIP step programs;
the classes here keep the reference's constructor/forward signatures for standalone use (inference, tests).
"""
import torch
from torch import nn
from torch.nn import functional as F

device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')

LOG_SIG_MAX = 2
LOG_SIG_MIN = -20


class _GaussianHead(nn.Module):
    """in -> hidden -> hidden (ReLU) -> (mean, clamped log_std)."""

    def __init__(self, input_dim, feature_dim, hidden_dim):
        super().__init__()
        self.l1 = nn.Linear(input_dim, hidden_dim)
        self.l2 = nn.Linear(hidden_dim, hidden_dim)
        self.mean_linear = nn.Linear(hidden_dim, feature_dim)
        self.log_std_linear = nn.Linear(hidden_dim, feature_dim)

    def _heads(self, x):
        h = F.relu(self.l2(F.relu(self.l1(x))))
        return self.mean_linear(h), self.log_std_linear(h).clamp(LOG_SIG_MIN, LOG_SIG_MAX)


class Encoder(_GaussianHead):
    """q(z | s, a, s')  (networks/vae.py:13-57)."""

    def __init__(self, state_dim, action_dim, feature_dim=256, hidden_dim=256):
        super().__init__(2 * state_dim + action_dim, feature_dim, hidden_dim)

    def forward(self, state, action, next_state):
        return self._heads(torch.cat([state, action, next_state], dim=-1))

    def sample(self, state, action, next_state):
        mean, log_std = self.forward(state, action, next_state)
        return mean + torch.randn_like(mean) * log_std.exp()


class Decoder(nn.Module):
    """z -> (s', r)  (networks/vae.py:60-86)."""

    def __init__(self, state_dim, feature_dim=256, hidden_dim=256):
        super().__init__()
        self.l1 = nn.Linear(feature_dim, hidden_dim)
        self.state_linear = nn.Linear(hidden_dim, state_dim)
        self.reward_linear = nn.Linear(hidden_dim, 1)

    def forward(self, feature):
        h = F.relu(self.l1(feature))
        return self.state_linear(h), self.reward_linear(h)


class GaussianFeature(_GaussianHead):
    """p(z | s, a)  (networks/vae.py:89-120)."""

    def __init__(self, state_dim, action_dim, feature_dim=256, hidden_dim=256):
        super().__init__(state_dim + action_dim, feature_dim, hidden_dim)

    def forward(self, state, action):
        return self._heads(torch.cat([state, action], dim=-1))
