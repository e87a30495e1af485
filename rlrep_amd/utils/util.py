"""Host utilities with the reference's names (utils/util.py:10-104).  No gym import: only its name was
used by the reference, and the agents here never needed it."""
import time
import numpy as np
import torch
from torch import nn


def unpack_batch(batch):
    # utils/util.py:10-11: note the order differs from the Batch field order
    return batch.state, batch.action, batch.next_state, batch.reward, batch.done


class Timer:
    """utils/util.py:14-37."""

    def __init__(self):
        self.reset()

    def reset(self):
        now = time.time()
        self._start_time, self._step_time, self._step = now, now, 0

    def set_step(self, step):
        self._step, self._step_time = step, time.time()

    def time_cost(self):
        return time.time() - self._start_time

    def steps_per_sec(self, step):
        now = time.time()
        sps = (step - self._step) / (now - self._step_time)
        self._step, self._step_time = step, now
        return sps


def eval_policy(policy, eval_env, eval_episodes=10):
    """utils/util.py:40-57 (pre-0.26 gym API: reset() -> obs, step() -> 4-tuple)."""
    total = 0.
    for _ in range(eval_episodes):
        state, done = eval_env.reset(), False
        while not done:
            state, reward, done, _ = eval_env.step(policy.select_action(np.array(state)))
            total += reward
    avg = total / eval_episodes
    print('---------------------------------------')
    print(f'Evaluation over {eval_episodes} episodes: {avg:.3f}')
    print('---------------------------------------')
    return avg


def weight_init(m):
    """utils/util.py:61-66: orthogonal weights, zero bias."""
    if isinstance(m, nn.Linear):
        nn.init.orthogonal_(m.weight.data)
        if hasattr(m.bias, 'data'):
            m.bias.data.fill_(0.0)


def mlp(input_dim, hidden_dim, output_dim, hidden_depth, output_mod=None):
    """utils/util.py:85-96: Linear(+ELU)*depth, Linear."""
    dims = [input_dim] + [hidden_dim] * hidden_depth
    mods = []
    for a, b in zip(dims[:-1], dims[1:]):
        mods += [nn.Linear(a, b), nn.ELU(inplace=True)]
    mods.append(nn.Linear(dims[-1], output_dim))
    if output_mod is not None:
        mods.append(output_mod)
    return nn.Sequential(*mods)


class MLP(nn.Module):
    """utils/util.py:69-82."""

    def __init__(self, input_dim, hidden_dim, output_dim, hidden_depth, output_mod=None):
        super().__init__()
        self.trunk = mlp(input_dim, hidden_dim, output_dim, hidden_depth, output_mod)
        self.apply(weight_init)

    def forward(self, x):
        return self.trunk(x)


def to_np(t):
    if t is None:
        return None
    if t.nelement() == 0:
        return np.array([])
    return t.cpu().detach().numpy()
