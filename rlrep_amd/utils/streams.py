"""Cheap forms of torch.cuda.current_stream() for the per-call host paths (no dependency on the HIP library)."""
import torch

# torch.cuda.current_stream() costs ~8 us of host time per call (device-index resolution, a Stream object): five of them sat in every iteration
# of main.py's loop.  The raw handle of the current stream is one C call; Stream OBJECTS (synchronize / wait_event / comparisons) are kept per handle.
try:
    from torch._C import _cuda_getCurrentRawStream as _raw_current_stream, _cuda_getDevice as _raw_current_device
except ImportError:                                     # (another torch build: the documented API)
    _raw_current_stream = _raw_current_device = None
_stream_objects = {}


def raw_stream():
    """hipStream_t of torch's current stream on the current device, as an int."""
    if _raw_current_stream is None:
        return torch.cuda.current_stream().cuda_stream
    return _raw_current_stream(_raw_current_device())


def current_stream():
    """torch.cuda.current_stream(), from a per-handle cache."""
    if _raw_current_stream is None:
        return torch.cuda.current_stream()
    di = _raw_current_device()
    key = (di, _raw_current_stream(di))
    s = _stream_objects.get(key)
    if s is None:
        s = _stream_objects[key] = torch.cuda.current_stream()
    return s


def current_device_index():
    """torch.cuda.current_device(), as one C call."""
    return _raw_current_device() if _raw_current_device is not None else torch.cuda.current_device()


class on_stream:
    """`with torch.cuda.stream(s):` for the per-call paths: the documented context manager resolves the device index twice on entry (~18 us of
    host time; three of them sat in every pipelined train()).  Same effect -- torch's current stream is `s` inside the block and restored after it
    -- through torch.cuda.set_stream and the cached current-stream lookup.  `s` must belong to the current device."""
    __slots__ = ('s', 'prev')

    def __init__(self, s):
        self.s = s

    def __enter__(self):
        self.prev = current_stream()
        torch.cuda.set_stream(self.s)
        return self.s

    def __exit__(self, *exc):
        torch.cuda.set_stream(self.prev)
        return False
