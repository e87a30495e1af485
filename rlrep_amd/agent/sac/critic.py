"""The SAC double-Q critic as a standalone torch module (module path and names of the reference's agent/sac/critic.py:15-36: DoubleQCritic) -- used
OUTSIDE the update path: tests, state_dict interchange.  Inside the agents the network lives in the flat parameter arenas and is evaluated by the HIP
step programs (csrc/gemm16.hip, csrc/elementwise.hip::qhead_*).  Attribute names (`Q1`, `Q2`) are kept so state_dicts match."""
import torch
from torch import nn

from rlrep_amd.utils.util import mlp, weight_init


class DoubleQCritic(nn.Module):
    HEADS = ('Q1', 'Q2')

    def __init__(self, obs_dim, action_dim, hidden_dim, hidden_depth):
        super().__init__()
        for name in self.HEADS:
            setattr(self, name, mlp(obs_dim + action_dim, hidden_dim, 1, hidden_depth))
        self.outputs = {}
        self.apply(weight_init)

    def forward(self, obs, action):
        if obs.shape[0] != action.shape[0]:
            raise AssertionError('obs / action batch mismatch')
        sa = torch.cat((obs, action), dim=-1)
        qs = tuple(getattr(self, name)(sa) for name in self.HEADS)
        self.outputs.update(q1=qs[0], q2=qs[1])
        return qs
