"""NumPy Pendulum-v1 with the pre-0.26 gym API the reference's main.py uses (`env.seed`, `reset() -> obs`,
`step() -> (obs, reward, done, info)`, `env._max_episode_steps`; main.py:52-56,111,133).

Dynamics are the public gym specification: g=10, m=l=1, dt=0.05, torque clipped to +-2,
angular velocity clipped to +-8, reward = -(wrap(theta)^2 + 0.1 thetadot^2 + 0.001 u^2), 200-step time limit,
reset theta ~ U(-pi, pi), thetadot ~ U(-1, 1).  Nothing here comes from the reference repository.
"""
import numpy as np


class Box:
    def __init__(self, low, high, rng):
        self.low, self.high = np.asarray(low, np.float32), np.asarray(high, np.float32)
        self.shape = self.low.shape
        self._rng = rng

    def sample(self):
        return self._rng.uniform(self.low, self.high).astype(np.float32)


class PendulumEnv:
    max_speed, max_torque, dt, g, m, l = 8.0, 2.0, 0.05, 10.0, 1.0, 1.0
    _max_episode_steps = 200

    def __init__(self, seed=None):
        self._rng = np.random.RandomState(seed)
        self.action_space = Box([-self.max_torque], [self.max_torque], self._rng)
        hi = np.array([1.0, 1.0, self.max_speed], np.float32)
        self.observation_space = Box(-hi, hi, self._rng)
        self._t = 0
        self._th, self._thd = 0.0, 0.0

    def seed(self, seed=None):
        self._rng.seed(seed)
        return [seed]

    def _obs(self):
        return np.array([np.cos(self._th), np.sin(self._th), self._thd], np.float32)

    def reset(self):
        self._th = self._rng.uniform(-np.pi, np.pi)
        self._thd = self._rng.uniform(-1.0, 1.0)
        self._t = 0
        return self._obs()

    def step(self, action):
        u = float(np.clip(np.asarray(action).reshape(-1)[0], -self.max_torque, self.max_torque))
        th, thd = self._th, self._thd
        wrapped = ((th + np.pi) % (2 * np.pi)) - np.pi
        cost = wrapped ** 2 + 0.1 * thd ** 2 + 0.001 * u ** 2
        thd = thd + (3 * self.g / (2 * self.l) * np.sin(th) + 3.0 / (self.m * self.l ** 2) * u) * self.dt
        thd = float(np.clip(thd, -self.max_speed, self.max_speed))
        th = th + thd * self.dt
        self._th, self._thd = th, thd
        self._t += 1
        return self._obs(), -cost, self._t >= self._max_episode_steps, {}


def make(name, seed=None):
    """`gym.make` stand-in: Pendulum-v1 natively; anything else through gym if it is installed."""
    if name.startswith('Pendulum'):
        return PendulumEnv(seed)
    try:
        import gym
    except ImportError as e:
        raise RuntimeError(f'environment {name!r} needs gym (not installed); only Pendulum-v1 is built in') from e
    return gym.make(name)
