"""One-shot gradient all-reduce over hipIpc-mapped inboxes (csrc/comm.hip).

The reference (haotiansun14/rl-rep) is a single process and has no collective; the data-parallel form of this package all-reduces the
gradient slice of every optimizer step (rlrep_amd/agent/sac/sac_agent.py `_allreduce`).  Those slices are 0.3 - 2 MB: latency-bound, the
wrong shape for a ring.  `OneShotAllReduce` is the latency-shaped form: every rank pushes its slice into its slot of every rank's inbox
(all xGMI links at once), signals, and adds the slots in rank order -- bit-identical sums on every rank, no float atomics.

OPT-IN (RLREP_ONESHOT_ALLREDUCE=1): it can be TESTED on a one-GPU box (several processes mapping each other's inbox on one device:
tests/test_comm.py) but only TIMED on a multi-GPU node, so RCCL stays the default.
"""
import ctypes as C

import torch

from ._lib import lib, check


class OneShotAllReduce:
    def __init__(self, max_floats, group=None):
        """Collective constructor: every rank of `group` (default: the world) calls it with the same `max_floats`.  The IPC handles travel
        through torch.distributed (any backend: gloo works)."""
        import torch.distributed as dist
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.max_floats = int(max_floats)
        h = C.c_void_p()
        torch.cuda.synchronize()
        check(lib.rlrep_comm_create(self.rank, self.world, self.max_floats, C.byref(h)), 'comm_create')
        self.h = h
        nb = lib.rlrep_comm_handle_bytes()
        mine = C.create_string_buffer(nb)
        check(lib.rlrep_comm_handle(self.h, mine, nb), 'comm_handle')
        handles = [None] * self.world
        dist.all_gather_object(handles, bytes(mine.raw), group=group)
        blob = C.create_string_buffer(b''.join(handles), nb * self.world)
        check(lib.rlrep_comm_connect(self.h, blob), 'comm_connect')
        dist.barrier(group=group)                # nobody pushes before every inbox is mapped everywhere
        self.fine_grained = bool(lib.rlrep_comm_fine_grained(self.h))

    def all_reduce(self, t, timeout_spins=0):
        """In-place sum over the ranks of a contiguous float32 CUDA tensor (<= max_floats elements, 16-byte aligned), stream-ordered on the
        current stream.  Every rank calls it with the same size in the same order."""
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() <= self.max_floats, (t.dtype, t.numel(), self.max_floats)
        check(lib.rlrep_comm_allreduce(self.h, C.c_void_p(t.data_ptr()), t.numel(), int(timeout_spins),
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'comm_allreduce')
        return t

    def check(self):
        """Synchronises the current stream; raises if a wait for a peer has timed out since the object was created."""
        m = C.c_uint32(0)
        check(lib.rlrep_comm_status(self.h, C.byref(m), C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'comm_status')

    def close(self):
        if getattr(self, 'h', None):
            lib.rlrep_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
