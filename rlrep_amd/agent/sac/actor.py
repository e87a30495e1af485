"""The SAC actor as a standalone torch module (module path and names of the reference's agent/sac/actor.py:16-91: TanhTransform,
SquashedNormal, DiagGaussianActor) -- used OUTSIDE the update path: inference, tests, state_dict interchange.  Inside the agents the network lives in
the flat parameter arenas and is evaluated by the HIP step programs (the policy epilogues of csrc/gemm16_tile.h, csrc/elementwise.hip::policy_fwd /
policy_bwd_kernel).  Attribute names (`trunk`) are kept so state_dicts match."""
import math

import torch
import torch.nn.functional as F
from torch import nn
from torch.distributions import Normal, TransformedDistribution, constraints
from torch.distributions.transforms import Transform

from rlrep_amd.utils.util import mlp, weight_init

_LOG2 = math.log(2.0)


class TanhTransform(Transform):
    """y = tanh(x); log|dy/dx| = 2 (log 2 - x - softplus(-2x)), the overflow-safe form."""
    domain, codomain = constraints.real, constraints.interval(-1.0, 1.0)
    bijective, sign = True, +1

    def __init__(self, cache_size=1):          # cache_size=1: log_prob reuses the pre-tanh sample, no atanh
        super().__init__(cache_size=cache_size)

    def __eq__(self, other):
        return type(other) is TanhTransform

    def _call(self, x):
        return torch.tanh(x)

    def _inverse(self, y):
        return torch.atanh(y)

    def log_abs_det_jacobian(self, x, y):
        return 2.0 * (_LOG2 - x - F.softplus(-2.0 * x))


class SquashedNormal(TransformedDistribution):
    def __init__(self, loc, scale):
        self.loc, self.scale = loc, scale
        self.base_dist = Normal(loc, scale)
        super().__init__(self.base_dist, [TanhTransform()])

    @property
    def mean(self):
        return torch.tanh(self.loc)


class DiagGaussianActor(nn.Module):
    def __init__(self, obs_dim, action_dim, hidden_dim, hidden_depth, log_std_bounds):
        super().__init__()
        self.log_std_bounds = tuple(log_std_bounds)
        self.trunk = mlp(obs_dim, hidden_dim, 2 * action_dim, hidden_depth)
        self.outputs = {}
        self.apply(weight_init)

    def forward(self, obs):
        lo, hi = self.log_std_bounds
        mu, raw = torch.chunk(self.trunk(obs), 2, dim=-1)
        log_std = lo + (hi - lo) * 0.5 * (1.0 + torch.tanh(raw))
        self.outputs.update(mu=mu, std=log_std.exp())
        return SquashedNormal(mu, self.outputs['std'])
