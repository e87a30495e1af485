"""rlrep_amd: MI355X-native (gfx950) update path of the rl-rep agents behind the reference's Python API.

Layout (mirrors the reference's module paths so its main.py-style launcher imports resolve):
    rlrep_amd.utils.{buffer,util}          <- utils/{buffer,util}.py
    rlrep_amd.agent.<alg>.<alg>_agent      <- agent/<alg>/<alg>_agent.py
    rlrep_amd.networks.{vae,critic,policy} <- networks/{vae,critic,policy}.py
    rlrep_amd.csrc / lib                   <- hand-written HIP kernels + C ABI (include/rlrep.h)
`import rlrep_amd.dropin` aliases these packages to the reference's top-level names (`utils`, `agent`,
`networks`).  (The directory is `rlrep_amd`, not `rl-rep_amd`: '-' is not valid in a Python identifier.)
"""
import os as _os

# HIP maps streams onto a few hardware queues (4 by default); streams that share one are serialised.  The pipelined train() needs its two
# launch chains (and, data parallel, one RCCL stream per process group) on different queues.  Only effective if the HIP runtime has not
# initialised yet, i.e. when this package is imported before the first GPU call of the process.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')

__version__ = '0.1.0'
