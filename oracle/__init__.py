"""CPU oracle for the rl-rep update hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, from scratch, the arithmetic of the reference's `agent.train()` path
(haotiansun14/rl-rep @ 2024-10-08) on PyTorch-CPU so that it can travel to the GPU box, where the
reference itself does not exist.  It is the *checker* for the hand-written HIP path:

  * only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it;
  * the product package (`rlrep_amd/`) never imports it and has no CPU fallback;
  * parity pinning: the reference holds no tests/golden vectors for this path,
    so the oracle is pinned against outputs of the reference itself, captured in this container by
    `tests/golden/make_fixtures.py` (reference imported unmodified, noise and indices recorded or
    injected) and committed as `tests/golden/*.npz`; `tests/test_oracle_golden.py` checks every one.

Gradients come from torch autograd (an independent derivation from the HIP path's hand-written
backward); Adam and Polyak are restated explicitly (`oracle/optim.py`).
"""
from .agents import make_oracle, Batch  # noqa: F401
