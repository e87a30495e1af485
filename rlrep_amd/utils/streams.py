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
