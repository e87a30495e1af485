"""Host-side helpers under the reference's names (utils/util.py:10-104): `unpack_batch`, `Timer`, `eval_policy`,
`weight_init`, `MLP`, `mlp`, `to_np`.  No gym import (the reference only needed the module's name)."""
import time

import numpy as np
import torch
from torch import nn


def unpack_batch(batch):
    """(state, action, next_state, reward, done) -- NOT the Batch field order (utils/util.py:10-11)."""
    return tuple(getattr(batch, f) for f in ('state', 'action', 'next_state', 'reward', 'done'))


class Timer:
    """Wall-clock bookkeeping for the launcher's `Steps per sec` line (utils/util.py:14-37)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self._start_time = self._step_time = time.time()
        self._step = 0

    def set_step(self, step):
        self._step, self._step_time = step, time.time()

    def time_cost(self):
        return time.time() - self._start_time

    def steps_per_sec(self, step):
        now = time.time()
        rate = (step - self._step) / max(now - self._step_time, 1e-12)
        self._step, self._step_time = step, now
        return rate


def eval_policy(policy, eval_env, eval_episodes=10):
    """Mean undiscounted return of the deterministic policy over `eval_episodes` episodes (pre-0.26 gym API)."""
    returns = []
    for _ in range(eval_episodes):
        obs, done, ret = eval_env.reset(), False, 0.0
        while not done:
            obs, reward, done, _ = eval_env.step(policy.select_action(np.asarray(obs)))
            ret += reward
        returns.append(ret)
    avg = float(np.mean(returns))
    bar = '-' * 39
    print(f'{bar}\nEvaluation over {eval_episodes} episodes: {avg:.3f}\n{bar}')
    return avg


def weight_init(m):
    """Orthogonal weights / zero bias for every nn.Linear (applied with Module.apply)."""
    if isinstance(m, nn.Linear):
        nn.init.orthogonal_(m.weight.data)
        if m.bias is not None:
            m.bias.data.zero_()


def mlp(input_dim, hidden_dim, output_dim, hidden_depth, output_mod=None):
    """nn.Sequential of `hidden_depth` x (Linear, ELU) + Linear; Linear layers sit at indices 0, 2, 4, ..."""
    widths = [input_dim] + [hidden_dim] * hidden_depth
    layers = []
    for fan_in, fan_out in zip(widths, widths[1:]):
        layers.extend((nn.Linear(fan_in, fan_out), nn.ELU(inplace=True)))
    layers.append(nn.Linear(widths[-1], output_dim))
    if output_mod is not None:
        layers.append(output_mod)
    return nn.Sequential(*layers)


class MLP(nn.Module):
    def __init__(self, input_dim, hidden_dim, output_dim, hidden_depth, output_mod=None):
        super().__init__()
        self.trunk = mlp(input_dim, hidden_dim, output_dim, hidden_depth, output_mod)
        self.apply(weight_init)

    def forward(self, x):
        return self.trunk(x)


def to_np(t):
    if t is None:
        return None
    return np.array([]) if t.nelement() == 0 else t.detach().cpu().numpy()
