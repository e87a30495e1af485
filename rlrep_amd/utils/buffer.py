"""Device-resident replay ring behind the reference's `ReplayBuffer` API (utils/buffer.py:13-48).

The reference keeps five float64 NumPy rings on the host and, on every `sample`, fancy-indexes them,
casts to float32 and issues five host->device copies.  Here the transitions live on the MI355X as ONE
fp32 array-of-structs ring, row = [state | action | next_state | reward | done] (2S+A+2 floats), so that
a minibatch is gathered by a single kernel (`rlrep_replay_sample`) and the concatenated network inputs
[s,a,s'] / [s,a] are prefixes of a row.  `add` stages rows in pinned host memory and flushes them with
one asynchronous copy per `sample`/`flush`.
"""
import collections
import numpy as np
import torch
from rlrep_amd.utils.streams import raw_stream as _raw_stream, current_stream as _current_stream, current_device_index as _current_device_index

Batch = collections.namedtuple('Batch', ['state', 'action', 'reward', 'next_state', 'done'])


class ReplayBuffer(object):
    def __init__(self, state_dim, action_dim, max_size=int(1e6), device=None, stage_rows=4096, shard=None):
        """shard=(rank, world): this ring is one shard of a replay partitioned by transition index: every rank is
        offered the same stream of transitions and keeps transition i iff i % world == rank, at slot i // world of its own ring (the
        default, shard=None, keeps everything: each data-parallel rank then owns the transitions of its own environment)."""
        self.max_size = int(max_size)
        self.shard = None if shard is None else (int(shard[0]), int(shard[1]))
        self._offered = 0
        self.ptr = 0
        self.size = 0
        self.state_dim, self.action_dim = int(state_dim), int(action_dim)
        self.row = 2 * self.state_dim + self.action_dim + 2
        self.device = torch.device(device if device is not None else
                                   ('cuda' if torch.cuda.is_available() else 'cpu'))
        self.ring = torch.zeros(self.max_size, self.row, dtype=torch.float32, device=self.device)
        pin = self.device.type == 'cuda'
        self._stage = torch.zeros(min(stage_rows, self.max_size), self.row, dtype=torch.float32, pin_memory=pin)
        self._stage_np = self._stage.numpy()
        self._staged = 0            # rows waiting in the staging buffer
        self._stage_start = 0       # ring position of the first staged row
        self._copy_done = None
        self._copy_event = None
        self._size_dev = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._size_pushed = -1
        self.device_epoch = 0        # bumped whenever this buffer enqueues device work on the caller's stream (a pipelined train() then orders itself after it)
        # hooks called before the ring / size scalar is overwritten on the caller's stream (a pipelined train() may still be sampling from
        # them on its own stream: each agent that trains from this buffer appends one that makes the caller's stream wait)
        self.before_device_write_hooks = []

    # ---- reference API ------------------------------------------------------------------------
    def add(self, state, action, next_state, reward, done):
        if self.shard is not None:
            i = self._offered
            self._offered += 1
            if i % self.shard[1] != self.shard[0]:
                return
        if self._staged == self._stage.shape[0]:
            self.flush()
        if self._copy_done is not None:
            self._copy_done.synchronize()
            self._copy_done = None
        if self._staged == 0:
            self._stage_start = self.ptr
        S, A = self.state_dim, self.action_dim
        r = self._stage_np[self._staged]
        r[:S] = state
        r[S:S + A] = action
        r[S + A:2 * S + A] = next_state
        r[2 * S + A] = reward
        r[2 * S + A + 1] = done
        self._staged += 1
        self.ptr = (self.ptr + 1) % self.max_size
        self.size = min(self.size + 1, self.max_size)

    def _before_device_write(self):
        for h in self.before_device_write_hooks:
            h()

    def flush(self):
        n = self._staged
        self._order_reads()          # (whoever samples after this call, on this stream, sees a flush another stream may still be performing)
        if n == 0:
            return
        a = self._stage_start
        first = min(n, self.max_size - a)
        self._before_device_write()
        if self.device.type == 'cuda':
            # the C ABI's entry for this row of the path (include/rlrep.h rlrep_replay_add): the staged rows, wrap-around included
            import ctypes as C
            from rlrep_amd._lib import lib, check
            # (one launch: the staged rows, read in place from the pinned staging buffer, and the new fill level for the device-side sampler;
            # the device context manager and a fresh Event per call were 15 us of host time in front of every train() of main.py's loop)
            def issue():
                check(lib.rlrep_replay_add_sized(C.c_void_p(self.ring.data_ptr()), self.max_size, self.row, a, C.c_void_p(self._stage.data_ptr()), n,
                                                 C.c_void_p(self._size_dev.data_ptr()), self.size, C.c_void_p(_raw_stream())), 'replay_add_sized')
                self._size_pushed = self.size
                if self._copy_event is None:
                    self._copy_event = torch.cuda.Event()
                self._copy_done = self._copy_event
                self._copy_done.record(_current_stream())
            if self.device.index is None or self.device.index == _current_device_index():
                issue()
            else:
                with torch.cuda.device(self.device):
                    issue()
        else:
            self.ring[a:a + first].copy_(self._stage[:first])
            if first < n:
                self.ring[:n - first].copy_(self._stage[first:n])
        self._staged = 0
        self.device_epoch += 1

    def load(self, state, action, next_state, reward, done):
        """Bulk-fill the ring (synthetic benchmarks / tests)."""
        if self.shard is not None:
            raise ValueError('load() fills the whole ring and knows nothing of the shard rule: add() the transitions instead')
        n = int(len(state))
        rows = np.concatenate([np.asarray(state, np.float32).reshape(n, -1), np.asarray(action, np.float32).reshape(n, -1),
                               np.asarray(next_state, np.float32).reshape(n, -1), np.asarray(reward, np.float32).reshape(n, 1),
                               np.asarray(done, np.float32).reshape(n, 1)], axis=1)
        self._before_device_write()
        self.ring[:n].copy_(torch.from_numpy(rows))
        self.size, self.ptr, self._staged = n, n % self.max_size, 0
        self.device_epoch += 1

    def save(self, path):
        self.flush()
        torch.cuda.synchronize() if self.device.type == 'cuda' else None
        np.savez_compressed(path, ring=self.ring[:self.size].cpu().numpy(), ptr=self.ptr, size=self.size, offered=self._offered)

    def restore(self, path):
        z = np.load(path)
        n = int(z['size'])
        self._before_device_write()
        self.ring[:n].copy_(torch.from_numpy(z['ring']))
        self.ptr, self.size, self._staged = int(z['ptr']), n, 0
        self._offered = int(z['offered']) if 'offered' in z.files else 0      # (files written before round 3 carry no shard counter)
        self.device_epoch += 1

    def size_dev(self):
        """int32[1] device scalar holding `size` (read by the graph-replayed index generator)."""
        if self._size_pushed != self.size:
            self._before_device_write()
            self._size_dev.fill_(self.size)
            self._size_pushed = self.size
            self.device_epoch += 1
        return self._size_dev

    def sample(self, batch_size):
        self.flush()
        ind = torch.from_numpy(np.random.randint(0, self.size, size=batch_size)).to(self.device)
        return self.gather(ind)

    def _order_reads(self):
        """A reader on the caller's stream: the last flush may have been issued on ANOTHER stream (a pipelined train() writes the staged rows on its
        feature stream, sac_agent.py) -- wait for it there."""
        if self._copy_done is not None and self.device.type == 'cuda':
            _current_stream().wait_event(self._copy_done)

    def gather(self, ind):
        S, A = self.state_dim, self.action_dim
        self._order_reads()
        rows = self.ring[ind.long()]
        return Batch(state=rows[:, :S].contiguous(), action=rows[:, S:S + A].contiguous(),
                     reward=rows[:, 2 * S + A:2 * S + A + 1].contiguous(),
                     next_state=rows[:, S + A:2 * S + A].contiguous(),
                     done=rows[:, 2 * S + A + 1:2 * S + A + 2].contiguous())

    # public array attributes of the reference, materialised on demand (read-only copies)
    def _col(self, a, b):
        self.flush()
        self._order_reads()
        return self.ring[:, a:b].cpu().numpy().astype(np.float64)

    @property
    def state(self):
        return self._col(0, self.state_dim)

    @property
    def action(self):
        return self._col(self.state_dim, self.state_dim + self.action_dim)

    @property
    def next_state(self):
        return self._col(self.state_dim + self.action_dim, 2 * self.state_dim + self.action_dim)

    @property
    def reward(self):
        return self._col(2 * self.state_dim + self.action_dim, 2 * self.state_dim + self.action_dim + 1)

    @property
    def done(self):
        return self._col(2 * self.state_dim + self.action_dim + 1, 2 * self.state_dim + self.action_dim + 2)
