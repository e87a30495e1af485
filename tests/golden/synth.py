"""Deterministic synthetic inputs shared by the fixture generator and the tests.

Everything here is derived from numpy's legacy ``RandomState`` (bit-stable across numpy versions),
so the config-dimension fixtures only have to store *outputs*: the initial parameters, the replay
content, the sample indices and every noise tensor can be regenerated on the GPU box.
"""
import numpy as np


def replay(S, A, n, seed=0):
    """state,next_state~N(0,1); action~U(-1,1); reward~N(0,1); done~Bern(0.01)."""
    rs = np.random.RandomState(seed)
    return dict(
        state=rs.randn(n, S).astype(np.float32),
        next_state=rs.randn(n, S).astype(np.float32),
        action=rs.uniform(-1, 1, (n, A)).astype(np.float32),
        reward=rs.randn(n, 1).astype(np.float32),
        done=(rs.uniform(size=(n, 1)) < 0.01).astype(np.float32),
    )


def init_like(shapes, seed=1234):
    """shapes: ordered list of (name, shape).  Weights/biases ~ U(-1/sqrt(fan_in), +) (nn.Linear-like
    scale), vlsac noise ~ N(0,1).  Returns an ordered dict name -> float32 array."""
    rs = np.random.RandomState(seed)
    out = {}
    fan = 1
    for name, shape in shapes:
        shape = tuple(shape)
        if name.endswith('noise'):
            out[name] = rs.standard_normal(shape).astype(np.float32)
            continue
        if name.endswith('weight'):
            fan = shape[-1]
        b = 1.0 / np.sqrt(max(fan, 1))
        out[name] = rs.uniform(-b, b, shape).astype(np.float32)
    return out


class NoiseSource:
    """Supplies sample indices and noise tensors in a fixed order from one RandomState."""

    def __init__(self, seed=4321):
        self.rs = np.random.RandomState(seed)

    def indices(self, hi, n):
        return self.rs.randint(0, hi, size=n).astype(np.int64)

    def normal(self, shape):
        return self.rs.standard_normal(tuple(shape)).astype(np.float32)
