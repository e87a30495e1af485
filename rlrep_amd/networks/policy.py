"""Module signature of the reference's networks/policy.py:13-94 (GaussianPolicy).  Imported by the reference's
vlsac agent but never constructed; the live policy is agent/sac/actor.py's tanh-squashed
Gaussian, which the HIP path implements in csrc/elementwise.hip (policy_fwd/bwd_kernel)."""
import torch
from torch import nn
from torch.nn import functional as F

LOG_SIG_MAX = 2
LOG_SIG_MIN = -20
epsilon = 1e-6

device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')


class GaussianPolicy(nn.Module):
    def __init__(self, state_dim, action_dim, action_space, hidden_dim=256):
        super().__init__()
        self.l1, self.l2 = nn.Linear(state_dim, hidden_dim), nn.Linear(hidden_dim, hidden_dim)
        self.mean_linear, self.log_std_linear = nn.Linear(hidden_dim, action_dim), nn.Linear(hidden_dim, action_dim)
        hi, lo = torch.as_tensor(action_space.high, dtype=torch.float32), torch.as_tensor(action_space.low, dtype=torch.float32)
        # plain tensor attributes on the module-level `device`, NOT buffers (policy.py:33-36): they are absent from state_dict()
        self.action_scale = ((hi - lo) / 2.).to(device)
        self.action_bias = ((hi + lo) / 2.).to(device)

    def forward(self, state):
        h = F.relu(self.l2(F.relu(self.l1(state))))
        return self.mean_linear(h), self.log_std_linear(h).clamp(LOG_SIG_MIN, LOG_SIG_MAX)

    def sample(self, state):
        """-> (action, log_prob[B,1], mean); log-prob correction log(1 - tanh^2 + 1e-6) (policy.py:90)."""
        mean, log_std = self.forward(state)
        std = log_std.exp()
        x = mean + torch.randn_like(mean) * std
        y = torch.tanh(x)
        log_prob = torch.distributions.Normal(mean, std).log_prob(x) - torch.log((1 - y.pow(2)) + epsilon)
        return (y * self.action_scale + self.action_bias, log_prob.sum(1, keepdim=True),
                torch.tanh(mean) * self.action_scale + self.action_bias)
