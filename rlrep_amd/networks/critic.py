"""Module signatures of the reference's networks/critic.py:6-148.  None of them has a live caller in the
reference; they are kept so imports and constructor/forward signatures resolve."""
import math
import torch
from torch import nn
from torch.nn import functional as F


class ValueCritic(nn.Module):
    def __init__(self, state_dim, hidden_dim=256):
        super().__init__()
        self.l1, self.l2, self.l3 = nn.Linear(state_dim, hidden_dim), nn.Linear(hidden_dim, hidden_dim), nn.Linear(hidden_dim, 1)

    def forward(self, state):
        return self.l3(F.relu(self.l2(F.relu(self.l1(state)))))


class _TwoHeads(nn.Module):
    """Two ReLU MLP heads l1-l3 / l4-l6 on a shared input of width `in_dim` with first-layer width `mid`."""

    def __init__(self, in_dim, mid, hidden_dim):
        super().__init__()
        self.l1, self.l2, self.l3 = nn.Linear(in_dim, mid), nn.Linear(mid, hidden_dim), nn.Linear(hidden_dim, 1)
        self.l4, self.l5, self.l6 = nn.Linear(in_dim, mid), nn.Linear(mid, hidden_dim), nn.Linear(hidden_dim, 1)

    def _q(self, x):
        q1 = self.l3(F.relu(self.l2(F.relu(self.l1(x)))))
        q2 = self.l6(F.relu(self.l5(F.relu(self.l4(x)))))
        return q1, q2


class Critic(_TwoHeads):
    def __init__(self, state_dim, action_dim, hidden_dim=256):
        super().__init__(state_dim + action_dim, hidden_dim, hidden_dim)

    def forward(self, state, action):
        return self._q(torch.cat([state, action], dim=-1))


class LinearCritic(_TwoHeads):
    def __init__(self, feature_dim, hidden_dim=256):
        super().__init__(feature_dim, hidden_dim, hidden_dim)

    def forward(self, x):
        return self._q(x)


class RFFLinearCritic(_TwoHeads):
    """Both first layers start from the SAME W ~ N(0,1), b ~ U(0, 2*3.1415926) (networks/critic.py:129-136);
    the activation is ReLU and `train_rff_weights` is ignored, as in the reference."""

    def __init__(self, feature_dim, num_rff=1024, hidden_dim=256, train_rff_weights=True):
        super().__init__(feature_dim, num_rff, hidden_dim)
        w = torch.randn_like(self.l1.weight)
        b = torch.rand_like(self.l1.bias) * 2 * 3.1415926
        with torch.no_grad():
            for lin in (self.l1, self.l4):
                lin.weight.copy_(w)
                lin.bias.copy_(b)

    def forward(self, x):
        return self._q(x)
