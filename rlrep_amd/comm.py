"""The shared gradient arena of the data-parallel form (csrc/comm.hip, csrc/dp_pull.h).

The reference (haotiansun14/rl-rep) is a single process and has no collective; the data-parallel form of this package sums the gradient slice
of every optimizer step over the ranks between the reference's `loss.backward()` and `optimizer.step()` (agent/vlsac/vlsac_agent.py:153-154,
183-184, 229-230 and siblings).  `GradientExchange` puts every rank's gradient arena into a block of device memory that all peers have mapped
over hipIpc; once `attach()`ed, the optimizer launches of the agent do the exchange themselves -- wait for the peers' gradients, sum every
rank's in rank order (bit-identical everywhere), do not end before every peer has read this rank's -- with NO extra launch and nothing for the
host to do between calls: a data-parallel train() is captured into the same hipGraphs as a single-GPU one.

`probe()` runs the same exchange as stand-alone launches on patterns whose rank-ordered sum is known and lets every rank agree on the outcome:
`HipCore` only attaches after a clean probe and falls back to torch.distributed all-reduces (RCCL) otherwise -- the exchange can be TESTED
on a one-GPU box (several processes mapping each other's block: tests/test_comm.py) but its first run across xGMI is the user's.
"""
import ctypes as C
import os
import socket

import numpy as np
import torch

from ._lib import lib, check


class _Arena:
    """A float32 device buffer owned by the library, seen by torch through __cuda_array_interface__ (no copy)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {'shape': (int(n),), 'typestr': '<f4', 'data': (int(ptr), False), 'version': 2, 'strides': None}


class GradientExchange:
    def __init__(self, arena_floats, group=None):
        """Collective constructor: every rank of `group` (default: the world) calls it with the same `arena_floats`.  The IPC handles and the
        placement of the ranks travel through torch.distributed (any backend: gloo works)."""
        import torch.distributed as dist
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.arena_floats = int(arena_floats)
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        # Every step that can fail on ONE rank only (allocation, export, mapping a peer's block) is caught and turned into a flag the ranks
        # agree on: a rank that raised here while its peers went on to the next collective would hang them.
        self.h, self.arena, self.fine_grained, self.error = None, None, False, None
        dev = torch.cuda.current_device()
        mine = None
        try:
            h = C.c_void_p()
            torch.cuda.synchronize()
            check(lib.rlrep_comm_create(self.rank, self.world, self.arena_floats, C.byref(h)), 'comm_create')
            self.h = h
            self.fine_grained = bool(lib.rlrep_comm_fine_grained(self.h))
            self.arena = torch.as_tensor(_Arena(lib.rlrep_comm_arena(self.h), self.arena_floats), device=f'cuda:{dev}')
            nb = lib.rlrep_comm_handle_bytes()
            buf = C.create_string_buffer(nb)
            check(lib.rlrep_comm_handle(self.h, buf, nb), 'comm_handle')
            mine = bytes(buf.raw)
        except Exception as e:                     # noqa: BLE001 (reported through self.error / probe())
            self.error = f'{type(e).__name__}: {e}'
        props = torch.cuda.get_device_properties(dev)
        where = (socket.gethostname(), str(getattr(props, 'uuid', '')) or str(getattr(props, 'pci_bus_id', dev)), dev)
        everyone = [None] * self.world
        dist.all_gather_object(everyone, (mine, where, self.fine_grained), group=group)
        self.same_device = len({w[1][:2] for w in everyone}) == 1
        # plain (coarse-grained) device memory is only coherent for peers on the SAME GPU; across GPUs the block must be fine-grained on every rank
        self.usable = all(w[0] is not None for w in everyone) and (self.same_device or all(w[2] for w in everyone))
        connected = False
        if self.usable:
            try:
                nb = lib.rlrep_comm_handle_bytes()
                blob = C.create_string_buffer(b''.join(w[0] for w in everyone), nb * self.world)
                check(lib.rlrep_comm_connect(self.h, blob), 'comm_connect')
                connected = True
            except Exception as e:                 # noqa: BLE001
                self.error = f'{type(e).__name__}: {e}'
        # nobody signals before every block is mapped everywhere -- and everybody learns whether it is
        self.usable = self._agree(self.usable and connected)
        self._scratch = None

    # ---- the stand-alone exchange (probe, tests) -------------------------------------------------------------------------------------
    def all_reduce(self, offset, n, out=None, timeout_spins=0):
        """out[0 .. n) = sum over the ranks, in rank order, of arena[offset .. offset + n): ONE launch on the current stream.  Every rank calls
        it with the same (offset, n) in the same order."""
        if out is None:
            out = torch.empty(int(n), dtype=torch.float32, device=self.arena.device)
        assert out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and out.numel() >= n
        check(lib.rlrep_comm_allreduce(self.h, int(offset), int(n), C.c_void_p(out.data_ptr()), int(timeout_spins),
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'comm_allreduce')
        return out

    def probe(self, rounds=3, n=None):
        """The exchange on known patterns, `rounds` times over the SAME addresses (a cache that served an earlier round's line would show), each
        compared with the rank-ordered float32 sum computed on the host.  Collective; returns True only if EVERY rank saw every round right and
        no wait ran out -- all ranks return the same value."""
        import torch.distributed as dist
        n = int(min(self.arena_floats, n or (1 << 18)))
        ok = self.usable
        if ok:
            try:
                for r in range(rounds):
                    pats = [self._pattern(q, r, n) for q in range(self.world)]
                    self.arena[:n].copy_(torch.from_numpy(pats[self.rank]))
                    torch.cuda.synchronize()
                    self._agree(True)                  # (every rank's pattern is in its arena before anybody reads it: the probe has no producer launch in front)
                    got = self.all_reduce(0, n).cpu().numpy()
                    want = pats[0].copy()
                    for q in range(1, self.world):
                        want = want + pats[q]
                    ok = ok and bool(np.array_equal(got, want))
                    if not ok and self.error is None:
                        self.error = f'probe round {r}: the exchange did not return the rank-ordered sum'
                ok = ok and self.status(raise_on_error=False) == 0
                self.arena[:n].zero_()
                torch.cuda.synchronize()
            except Exception as e:                     # noqa: BLE001 (a rank that raised here would leave its peers in the next collective)
                ok, self.error = False, f'{type(e).__name__}: {e}'
        return self._agree(ok)

    def _agree(self, ok):
        """True only if `ok` on every rank (a collective; also the barrier between mapping and first use)."""
        import torch.distributed as dist
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
        if dist.get_backend(self.group) == 'nccl':
            flag = flag.cuda()
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(flag.item()))

    @staticmethod
    def _pattern(rank, rnd, n):
        rs = np.random.RandomState(7919 * rnd + rank + 1)
        return (rs.standard_normal(n) * (10.0 ** rs.randint(-3, 4, size=n))).astype(np.float32)

    # ---- attachment and status ----------------------------------------------------------------------------------------------------------
    def attach(self, agent_handle, max_floats):
        """-> set of optimizer groups whose launches now carry the exchange (gradient slices of at most max_floats floats)."""
        mask = C.c_int32(0)
        check(lib.rlrep_comm_attach(agent_handle, self.h, int(max_floats), C.byref(mask)), 'comm_attach')
        return {g for g in range(4) if mask.value & (1 << g)}

    def status(self, raise_on_error=True, clear=False):
        """Late-rank mask of the waits that ran out so far (0: none).  Reads a word in mapped host memory: no device synchronisation."""
        if not self.h:
            return 0
        m = C.c_uint32(0)
        rc = lib.rlrep_comm_status(self.h, C.byref(m), 1 if clear else 0)
        if rc != 0 and raise_on_error:
            check(rc, 'comm_status')
        return int(m.value)

    def close(self):
        if getattr(self, 'h', None):
            self.arena = None
            lib.rlrep_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
