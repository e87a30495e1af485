"""Standalone module with the reference's agent/sac/critic.py name (DoubleQCritic, critic.py:15-36)."""
import torch
from torch import nn

from rlrep_amd.utils import util


class DoubleQCritic(nn.Module):
    def __init__(self, obs_dim, action_dim, hidden_dim, hidden_depth):
        super().__init__()
        self.Q1 = util.mlp(obs_dim + action_dim, hidden_dim, 1, hidden_depth)
        self.Q2 = util.mlp(obs_dim + action_dim, hidden_dim, 1, hidden_depth)
        self.outputs = dict()
        self.apply(util.weight_init)

    def forward(self, obs, action):
        assert obs.size(0) == action.size(0)
        x = torch.cat([obs, action], dim=-1)
        self.outputs['q1'], self.outputs['q2'] = self.Q1(x), self.Q2(x)
        return self.outputs['q1'], self.outputs['q2']
