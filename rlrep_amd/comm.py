"""The shared gradient arena of the data-parallel form (csrc/comm.hip, csrc/dp_pull.h).

The reference (haotiansun14/rl-rep) is a single process and has no collective; the data-parallel form of this package sums the gradient slice
of every optimizer step over the ranks between the reference's `loss.backward()` and `optimizer.step()` (agent/vlsac/vlsac_agent.py:153-154,
183-184, 229-230 and siblings).  `GradientExchange` puts every rank's gradient arena into a block of device memory that all peers have mapped
over hipIpc; once `attach()`ed, the optimizer launches of the agent do the exchange themselves -- wait for the peers' gradients, sum every
rank's in rank order (bit-identical everywhere), do not end before every peer has read this rank's -- with NO extra launch and nothing for the
host to do between calls: a data-parallel train() is captured into the same hipGraphs as a single-GPU one.

`probe()` runs the same exchanges as stand-alone launches on patterns whose rank-ordered sum is known -- first filled by the host, then by a
PRODUCER KERNEL that is the graph node right in front of the pull (the production order: no host between the launch that writes gradients and
the launch that signals READY), one-shot and two-shot, plus the pushed-slot exchange -- and lets every rank agree on the outcome: `HipCore` only
attaches after a clean probe and falls back to torch.distributed all-reduces (RCCL) otherwise -- the exchange can be TESTED on a one-GPU box
(several processes mapping each other's block: tests/test_comm.py; several ranks in one process: `LoopbackGroup`) but its first run across
xGMI is the user's.
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


class _ExchangeBase:
    """What both forms share: the stand-alone exchanges, attachment, the error word."""
    h = None
    arena = None
    scratch_floats = 0

    def all_reduce(self, offset, n, out=None, mode=0, timeout_us=0):
        """out[0 .. n) = sum over the ranks, in rank order, of block[offset .. offset + n): ONE launch on the current stream (mode 1: one-shot
        pull, 2: two-shot, 0: by size).  Every rank calls it with the same (offset, n, mode) in the same order."""
        if out is None:
            out = torch.empty(int(n), dtype=torch.float32, device=self.arena.device)
        assert out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and out.numel() >= n
        check(lib.rlrep_comm_allreduce(self.h, int(offset), int(n), C.c_void_p(out.data_ptr()), int(mode), int(timeout_us),
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'comm_allreduce')
        return out

    def all_gather(self, offset, n):
        """block[offset + q n .. + n) of every rank q into every rank's block (one pull launch on the current stream)."""
        check(lib.rlrep_comm_allgather(self.h, int(offset), int(n), C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'comm_allgather')

    def probe_slots(self, n, rnd, out=None, timeout_us=0):
        """The pushed-slot exchange on the pattern of round `rnd` (two launches on the current stream) -> out[0 .. n) = rank-ordered sum."""
        if out is None:
            out = torch.empty(int(n), dtype=torch.float32, device=self.arena.device)
        check(lib.rlrep_comm_probe_slots(self.h, int(n), int(rnd), C.c_void_p(out.data_ptr()), int(timeout_us), C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'comm_probe_slots')
        return out

    def probe_fill(self, offset, n, rnd):
        check(lib.rlrep_comm_probe_fill(self.h, int(offset), int(n), int(rnd), C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'comm_probe_fill')

    @staticmethod
    def probe_pattern(rank, rnd, n):
        """What probe_fill writes (the same function evaluated on the host: csrc/comm.hip comm_pattern)."""
        i = np.arange(n, dtype=np.uint64)
        h = ((i * np.uint64(2654435761)) & np.uint64(0xffffffff)).astype(np.uint32) ^ np.uint32((rank * 40503 + rnd * 9176 + 12345) & 0xffffffff)
        h ^= h >> np.uint32(15)
        h = (h.astype(np.uint64) * np.uint64(2246822519) & np.uint64(0xffffffff)).astype(np.uint32)
        h ^= h >> np.uint32(13)
        m = (h & np.uint32(0xffff)).astype(np.int32).astype(np.float32) * np.float32(1.0 / 32768.0) - np.float32(1.0)
        ex = ((h >> np.uint32(16)) % np.uint32(13)).astype(np.int32) - 6
        return np.ldexp(m, ex).astype(np.float32)

    def set_timeout(self, seconds):
        """Bound of every device-side wait of attachments made from now on (default 120 s: a watchdog, not a schedule)."""
        check(lib.rlrep_comm_set_timeout(self.h, int(float(seconds) * 1e6)), 'comm_set_timeout')

    def attach(self, agent_handle, max_floats, two_shot_floats=0):
        """-> set of optimizer groups whose launches now carry the exchange (gradient slices of at most max_floats floats; slices of at least
        two_shot_floats take the two-shot form when world >= 3).  Rebuilds the agent's step programs: call before capturing graphs."""
        mask = C.c_int32(0)
        check(lib.rlrep_comm_attach(agent_handle, self.h, int(max_floats), int(two_shot_floats), C.byref(mask)), 'comm_attach')
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


class GradientExchange(_ExchangeBase):
    def __init__(self, arena_floats, scratch_floats=0, group=None):
        """Collective constructor: every rank of `group` (default: the world) calls it with the same sizes.  The IPC handles and the
        placement of the ranks travel through torch.distributed (any backend: gloo works)."""
        import torch.distributed as dist
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.arena_floats, self.scratch_floats = int(arena_floats), int(scratch_floats)
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        # Every step that can fail on ONE rank only (allocation, export, mapping a peer's block) is caught and turned into a flag the ranks
        # agree on: a rank that raised here while its peers went on to the next collective would hang them.
        self.h, self.arena, self.fine_grained, self.error = None, None, False, None
        dev = torch.cuda.current_device()
        mine = None
        try:
            h = C.c_void_p()
            torch.cuda.synchronize()
            check(lib.rlrep_comm_create(self.rank, self.world, self.arena_floats, self.scratch_floats, C.byref(h)), 'comm_create')
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

    # ---- the probe ------------------------------------------------------------------------------------------------------------------------
    def probe(self, rounds=3, n=None):
        """The exchanges on known patterns, each compared with the rank-ordered float32 sum computed on the host:
          (1) `rounds` times over the SAME addresses with the arena filled by the host (a cache that served an earlier round's line would show);
          (2) ONE hipGraph of `rounds` x [producer kernel -> pull], replayed twice: the pattern is written by an ordinary launch with ordinary
              stores and the NEXT node signals READY and pulls -- no host synchronisation in between, as in a train() -- one-shot and (world >= 3)
              two-shot; the next round's producer overwrites the arena right behind the pull (the DONE handshake);
        Collective with a FIXED sequence of collectives whatever happens on any rank (a rank that raises still takes part in every agreement,
        so no rank can be left waiting in one); returns True only if EVERY rank saw everything right and no wait ran out."""
        n = int(min(self.arena_floats, n or (1 << 18))) & ~3
        ok = self.usable
        tmo = 10_000_000        # every wait of the probe is bounded by ten seconds (a node where it fails must fall back within the minute)

        def step(fn):
            """run fn on every rank that is still good; all ranks then agree (one collective per step, always executed)"""
            nonlocal ok
            if ok:
                try:
                    res = fn()
                    if res is False:
                        ok = False
                except Exception as e:             # noqa: BLE001
                    ok, self.error = False, self.error or f'{type(e).__name__}: {e}'
            ok = self._agree(ok)
            return ok

        def want(rnd, pat):
            w = pat(0, rnd, n).copy()
            for q in range(1, self.world):
                w = w + pat(q, rnd, n)
            return w

        for r in range(rounds):
            def fill(r=r):
                self.arena[:n].copy_(torch.from_numpy(self._pattern(self.rank, r, n)))
                torch.cuda.synchronize()
            step(fill)                                 # (every rank's pattern is in its arena before anybody reads it: this form has no producer launch in front)

            def pull(r=r):
                got = self.all_reduce(0, n, mode=1, timeout_us=tmo).cpu().numpy()
                if not np.array_equal(got, want(r, self._pattern)):
                    self.error = self.error or f'probe round {r}: the exchange did not return the rank-ordered sum (host-filled arena)'
                    return False
            step(pull)
        for mode in ((1, 2) if self.world >= 3 else (1,)):
            outs = []

            def build(mode=mode):
                outs[:] = [torch.zeros(n, dtype=torch.float32, device=self.arena.device) for _ in range(rounds)]
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    g.capture_begin(capture_error_mode='thread_local')
                    try:
                        for r in range(rounds):
                            self.probe_fill(0, n, 100 * mode + r)          # producer: an ordinary kernel writes this rank's pattern ...
                            self.all_reduce(0, n, out=outs[r], mode=mode, timeout_us=tmo)  # ... and the NEXT node pulls (no host in between)
                    finally:
                        g.capture_end()
                torch.cuda.current_stream().wait_stream(s)
                self._probe_graph = g
            step(build)

            def replay(mode=mode):
                for rep in range(2):
                    self._probe_graph.replay()
                torch.cuda.synchronize()
                for r in range(rounds):
                    if not np.array_equal(outs[r].cpu().numpy(), want(100 * mode + r, self.probe_pattern)):
                        self.error = self.error or f'probe: producer kernel -> pull in one graph (mode {mode}, round {r}) did not return the rank-ordered sum'
                        return False
            step(replay)
            self._probe_graph = None

        ns = min(512, self.scratch_floats // (2 * self.world)) if self.scratch_floats else 0
        if ns >= 64:
            # (3) the PUSHED exchange (spedersac's Phibar / v): producer launch stores into every rank's slots, the next launch sums its own block's
            def slots():
                outs = [self.probe_slots(ns, 300 + r, timeout_us=tmo) for r in range(rounds + 1)]      # (rounds + 1: both parities twice)
                torch.cuda.synchronize()
                for r, o in enumerate(outs):
                    w = self.probe_pattern(0, 300 + r, ns).copy()
                    for q in range(1, self.world):
                        w = w + self.probe_pattern(q, 300 + r, ns)
                    if not np.array_equal(o.cpu().numpy(), w):
                        self.error = self.error or f'probe: pushed-slot exchange, round {r}: not the rank-ordered sum'
                        return False
            step(slots)

        def finish():
            if self.status(raise_on_error=False) != 0:
                self.error = self.error or 'probe: a wait ran out'
                return False
            self.arena[:n].zero_()
            torch.cuda.synchronize()
        step(finish)
        return ok

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


class LoopbackGroup:
    """`world` ranks inside ONE process on one GPU: every rank's block is plain device memory and the peers are plain pointers
    (rlrep_comm_connect_local) -- no IPC, no torch.distributed, no process time-slicing.  What tools/exp/dp_loopback.py measures the cost of the
    in-launch exchange with (N agents on N stream pairs, attached against unattached) and what the one-process multi-rank tests run on.
    Launches of different ranks must be issued on DIFFERENT streams (a rank's optimizer launch waits for the peers' on the device), and the
    host must never wait for one rank before it has issued the matching launches of the others."""

    def __init__(self, world, arena_floats=None, scratch_floats=0):
        """arena_floats None: the blocks are allocated when the first agent joins (HipCore calls ensure() with its layout's sizes)."""
        self.world, self.arena_floats, self.scratch_floats = int(world), None, 0
        self.members = []
        self.reference_core = None          # rank 0's HipCore: the later ranks copy their initial parameters from it
        self.timeout_s = None
        if arena_floats is not None:
            self.ensure(arena_floats, scratch_floats)

    def ensure(self, arena_floats, scratch_floats=0):
        if self.members:
            assert self.arena_floats >= int(arena_floats) and self.scratch_floats >= int(scratch_floats), 'loopback group is too small for this agent'
            return
        self.arena_floats, self.scratch_floats = int(arena_floats), int(scratch_floats)
        self.members = [LoopbackExchange(self, r) for r in range(self.world)]
        arr = (C.c_void_p * self.world)(*[m.h for m in self.members])
        for m in self.members:
            check(lib.rlrep_comm_connect_local(m.h, arr), 'comm_connect_local')
            if self.timeout_s is not None:
                m.set_timeout(self.timeout_s)

    def __getitem__(self, rank):
        return self.members[rank]

    def close(self):
        torch.cuda.synchronize()
        for m in self.members:
            m.close()


class LoopbackExchange(_ExchangeBase):
    def __init__(self, group, rank):
        self.group, self.rank, self.world = group, int(rank), group.world
        self.arena_floats, self.scratch_floats = group.arena_floats, group.scratch_floats
        self.fine_grained, self.same_device, self.usable, self.error = False, True, True, None
        h = C.c_void_p()
        torch.cuda.synchronize()
        check(lib.rlrep_comm_create(self.rank, self.world, self.arena_floats, self.scratch_floats, C.byref(h)), 'comm_create')
        self.h = h
        self.arena = torch.as_tensor(_Arena(lib.rlrep_comm_arena(self.h), self.arena_floats), device=f'cuda:{torch.cuda.current_device()}')

    def probe(self, *a, **k):
        return True                         # (same device, plain pointers: nothing to find out; tests/test_comm.py checks the arithmetic)


_CONCURRENT = {}


def concurrent_streams(n, tries=48):
    """`n` torch streams of the current device that really run side by side.  HIP maps streams onto a few hardware queues and two streams
    that share one are serialised; a launch that WAITS on the device for a launch queued behind it in the same queue never ends.  Found by
    trial, not by timing: a two-rank exchange (one rank per stream, 100 ms bound) completes only if the two streams are concurrent.  The set
    is kept for the life of the process; raises if the runtime does not have `n` concurrent queues (this image: four per process)."""
    dev = torch.cuda.current_device()
    kept = _CONCURRENT.setdefault(dev, [])
    if len(kept) >= n:
        return kept[:n]
    grp = LoopbackGroup(2, 64)
    try:
        def together(a, b):
            for first, second in ((a, b), (b, a)):
                with torch.cuda.stream(first):
                    grp[0].all_reduce(0, 4, mode=1, timeout_us=100000)
                with torch.cuda.stream(second):
                    grp[1].all_reduce(0, 4, mode=1, timeout_us=100000)
                torch.cuda.synchronize()
                bad = grp[0].status(raise_on_error=False, clear=True) | grp[1].status(raise_on_error=False, clear=True)
                if bad:
                    return False
            return True
        # (first launches load code: run the exchange once -- one rank after the other, its one-millisecond waits running out -- before anything is judged)
        for r in (0, 1):
            grp[r].all_reduce(0, 4, mode=1, timeout_us=1000)
            torch.cuda.synchronize()
        for r in (0, 1):
            grp[r].status(raise_on_error=False, clear=True)
        if not kept:
            kept.append(torch.cuda.Stream())
        while len(kept) < n and tries > 0:
            tries -= 1
            cand = torch.cuda.Stream()
            if all(together(k, cand) for k in kept):
                kept.append(cand)
    finally:
        grp.close()
    if len(kept) < n:
        raise RuntimeError(f'rlrep_amd: found only {len(kept)} mutually concurrent HIP streams, {n} are needed (hardware queues per process: GPU_MAX_HW_QUEUES)')
    return kept[:n]
