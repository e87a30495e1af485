"""Standalone modules with the reference's agent/sac/actor.py names (TanhTransform, SquashedNormal,
DiagGaussianActor).  Inside the agents the actor lives in the parameter arena and runs in
csrc/elementwise.hip::policy_fwd/bwd_kernel; these classes serve inference/tests outside the update path."""
import math
import torch
from torch import nn
import torch.nn.functional as F
from torch import distributions as pyd

from rlrep_amd.utils import util


class TanhTransform(pyd.transforms.Transform):
    """y = tanh x with the numerically stable log|dy/dx| = 2 (log 2 - x - softplus(-2x))  (actor.py:16-43)."""
    domain = pyd.constraints.real
    codomain = pyd.constraints.interval(-1.0, 1.0)
    bijective = True
    sign = +1

    def __init__(self, cache_size=1):
        super().__init__(cache_size=cache_size)

    def __eq__(self, other):
        return isinstance(other, TanhTransform)

    def _call(self, x):
        return torch.tanh(x)

    def _inverse(self, y):
        return 0.5 * (torch.log1p(y) - torch.log1p(-y))

    def log_abs_det_jacobian(self, x, y):
        return 2.0 * (math.log(2.0) - x - F.softplus(-2.0 * x))


class SquashedNormal(pyd.transformed_distribution.TransformedDistribution):
    """tanh-squashed diagonal Gaussian; `.mean` is tanh(loc)  (actor.py:46-60)."""

    def __init__(self, loc, scale):
        self.loc, self.scale = loc, scale
        self.base_dist = pyd.Normal(loc, scale)
        super().__init__(self.base_dist, [TanhTransform()])

    @property
    def mean(self):
        return torch.tanh(self.loc)


class DiagGaussianActor(nn.Module):
    """obs -> SquashedNormal; log_std = lo + (hi-lo)/2 * (tanh(raw)+1)  (actor.py:63-91)."""

    def __init__(self, obs_dim, action_dim, hidden_dim, hidden_depth, log_std_bounds):
        super().__init__()
        self.log_std_bounds = log_std_bounds
        self.trunk = util.mlp(obs_dim, hidden_dim, 2 * action_dim, hidden_depth)
        self.outputs = dict()
        self.apply(util.weight_init)

    def forward(self, obs):
        mu, raw = self.trunk(obs).chunk(2, dim=-1)
        lo, hi = self.log_std_bounds
        std = (lo + 0.5 * (hi - lo) * (torch.tanh(raw) + 1)).exp()
        self.outputs['mu'], self.outputs['std'] = mu, std
        return SquashedNormal(mu, std)
