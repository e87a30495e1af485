"""Module path of the reference's agent/sac/critic.py; the class lives in rlrep_amd/agent/sac/modules.py."""
from rlrep_amd.agent.sac.modules import DoubleQCritic  # noqa: F401
