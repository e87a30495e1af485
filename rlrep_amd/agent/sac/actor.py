"""Module path of the reference's agent/sac/actor.py; the classes live in rlrep_amd/agent/sac/modules.py."""
from rlrep_amd.agent.sac.modules import TanhTransform, SquashedNormal, DiagGaussianActor  # noqa: F401
