"""Alias rlrep_amd's subpackages to the reference's top-level module names.

    import rlrep_amd.dropin            # before the reference-style launcher's own imports
    from utils import util, buffer     # -> rlrep_amd.utils
    from agent.vlsac import vlsac_agent
"""
import importlib
import sys

for _name in ('utils', 'utils.util', 'utils.buffer', 'networks', 'networks.vae', 'networks.critic',
              'networks.policy', 'agent', 'agent.sac', 'agent.sac.sac_agent', 'agent.sac.actor', 'agent.sac.critic',
              'agent.vlsac', 'agent.vlsac.vlsac_agent', 'agent.ctrlsac', 'agent.ctrlsac.ctrlsac_agent',
              'agent.spedersac', 'agent.spedersac.spedersac_agent', 'agent.diffsrsac',
              'agent.diffsrsac.diffsrsac_agent'):
    try:
        sys.modules[_name] = importlib.import_module('rlrep_amd.' + _name)
    except ModuleNotFoundError:
        pass
