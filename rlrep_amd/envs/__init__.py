"""Minimal environments so that BASELINE config (1) (sac on Pendulum-v1) runs with no gym/MuJoCo installed."""
from .pendulum import PendulumEnv, make  # noqa: F401
