"""HipCore: owns the device arenas of one agent and drives the step programs of librlrep_hip.so.

torch is used for what the C ABI asks the caller to provide -- device memory (arenas, workspace, noise
and index buffers), the current HIP stream -- and nothing else: no torch op runs on the update path.
"""
import ctypes as C
import os
import warnings
import weakref
import numpy as np
import torch

from . import _lib
from ._lib import lib, check

METRIC_SLOTS = 16


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


from rlrep_amd.utils.streams import raw_stream, current_stream          # (cheap forms of torch.cuda.current_stream())


def _stream():
    return C.c_void_p(raw_stream())


class LazyInfo(dict):
    """The dict `train()` returns.  The reference calls `.item()` 4-25 times per train() (SURVEY quirk
    Q14), each a device sync; here the metric slots are snapshotted on the device and only fetched when a
    value is actually read (main.py reads `info` once per 5000 steps)."""

    def __init__(self, names, snapshot, early=None, after=None, on_read=None):
        """early = (keys, callable): a subset of the metrics that is final BEFORE the full snapshot is (pipelined train(): the feature
        steps' losses are final when the feature chain ends, while the critic / actor chain is still running); reading only such keys
        fetches them through `callable` and does not wait for the rest."""
        super().__init__()
        self._names, self._snap, self._done = names, snapshot, False
        self._after = after          # called once the values are on the host (HipCore.chain_check: the device-side checks of the chain launches)
        self._on_read = on_read      # called when the CALLER reads the dict for the first time (not when the library resolves it early)
        self._early_keys, self._early_snap, self._early_done = (frozenset(early[0]), early[1], False) if early else (frozenset(), None, False)
        for n in names:
            if n:
                dict.__setitem__(self, n, None)

    # quirk Q14: the reference returns these two as 0-dim TENSORS (sac_agent.py:163-166: `alpha_loss` fp32, `alpha` = log_alpha.exp() fp64),
    # everything else as Python floats
    TENSOR_KEYS = {'alpha_loss': torch.float32, 'alpha': torch.float64}

    def _fetch(self, host_ring=None):
        """host_ring: a host copy of the metric history ring (HipCore.history_resolve: the library resolves unread dicts before the ring
        wraps); None: the caller is reading."""
        if not self._done:
            if host_ring is None and self._on_read is not None:
                self._on_read()
            if host_ring is not None:
                snap = self._snap(host_ring)
            else:
                snap = self._snap() if callable(self._snap) else self._snap     # callable: fetched (and flushed) on first read
            vals = snap.cpu().numpy()
            if self._after is not None and host_ring is None:
                self._after()
            for i, n in enumerate(self._names):
                if n:
                    v = float(vals[i])
                    dict.__setitem__(self, n, torch.tensor(v, dtype=self.TENSOR_KEYS[n]) if n in self.TENSOR_KEYS else v)
            self._done = True

    def _fetch_early(self):
        if not (self._done or self._early_done):
            vals = self._early_snap().cpu().numpy()
            for i, n in enumerate(self._names):
                if n in self._early_keys:
                    dict.__setitem__(self, n, float(vals[i]))
            self._early_done = True

    def __getitem__(self, k):
        if k in self._early_keys and not self._done:
            self._fetch_early()
        else:
            self._fetch()
        return dict.__getitem__(self, k)

    def get(self, k, d=None):
        self._fetch()
        return dict.get(self, k, d)

    def items(self):
        self._fetch()
        return dict.items(self)

    def values(self):
        self._fetch()
        return dict.values(self)

    def __repr__(self):
        self._fetch()
        return dict.__repr__(self)


class HipCore:
    def __init__(self, alg, dims, hyper, device=None, world_size=1, loopback=None):
        """loopback: a (LoopbackGroup, rank) pair -- this core is rank `rank` of `world_size` replicas inside ONE process (rlrep_amd/comm.py)."""
        if not torch.cuda.is_available():
            raise RuntimeError('rlrep_amd needs an MI355X (no CPU fallback): torch.cuda.is_available() is False')
        self.alg = alg
        self.device = torch.device(device if device is not None else f'cuda:{torch.cuda.current_device()}')
        self.dims = _lib.Dims()
        self.dims.alg = _lib.ALG[alg]
        for k, v in dims.items():
            setattr(self.dims, k, int(v))
        self.hyper = _lib.Hyper()
        for k, v in hyper.items():
            setattr(self.hyper, k, v)
        self.hyper.world_size = int(world_size)
        self.dims.world_size = int(world_size)          # the workspace is sized for the data-parallel forms of the feature step (ctrlsac: [B, world * B] scores)
        self.hyper.beta1, self.hyper.beta2, self.hyper.adam_eps = 0.9, 0.999, 1e-8
        info = _lib.LayoutInfo()
        check(lib.rlrep_layout(C.byref(self.dims), C.byref(info), None, 0), 'layout')
        descs = (_lib.TensorDesc * info.n_tensors)()
        check(lib.rlrep_layout(C.byref(self.dims), C.byref(info), descs, info.n_tensors), 'layout')
        self.layout = info
        self.descs = {d.name.decode(): (d.arena, d.group, d.offset, d.rows, d.cols) for d in descs}
        self.order = [d.name.decode() for d in descs]
        dev = self.device
        # ONE device allocation, the seven arenas carved out of it at 256-byte boundaries: the tile engine's fast front ends address every
        # operand of a launch as a 32-bit float offset from one base (csrc/gemm16.hip fast_args), so whether a launch qualifies must depend
        # on its shapes, never on where the allocator happened to put two separately allocated tensors (activations live in the workspace,
        # weights in the parameter arena).  The whole block must span < 16 GiB for that; Humanoid's is 1.7 GB.
        sizes = [4 * info.param_floats, 4 * info.target_floats, 4 * info.grad_floats, 4 * info.param_floats, 4 * info.param_floats,
                 int(info.workspace_bytes), 4 * 8]
        offs, total = [], 0
        for n in sizes:
            offs.append(total)
            total += (int(n) + 255) & ~255
        if total >= (1 << 34):
            raise RuntimeError(f'rlrep_amd: the arenas of this agent need {total} bytes in one block; the tile engine addresses operands '
                               'as 32-bit float offsets from one base (16 GiB)')
        self._block = torch.zeros(total + 256, dtype=torch.uint8, device=dev)
        skew = (-self._block.data_ptr()) & 255

        def carve(i, dtype):
            return self._block[skew + offs[i]:skew + offs[i] + sizes[i]].view(dtype)
        self.params, self.targets, self.grads = carve(0, torch.float32), carve(1, torch.float32), carve(2, torch.float32)
        # Data parallel: the gradient arena moves into a block every peer has mapped (rlrep_amd/comm.py), and the optimizer launches sum the
        # ranks' gradients themselves -- but only after the exchange has passed its probe on THIS set of ranks (same answer on every rank);
        # otherwise the arena stays where it is and the agent all-reduces with torch.distributed between backward and apply, as before.
        self.exchange, self.fused_groups = None, frozenset()
        if loopback is not None:
            group, lrank = loopback
            group.ensure(info.grad_floats, info.exchange_floats)
            assert group.world == int(world_size) and self.dims.rank == lrank, 'loopback group does not fit this agent'
            self.exchange = group[lrank]
            self.grads = self.exchange.arena[:info.grad_floats]
        elif int(world_size) > 1 and os.environ.get('RLREP_DP_FUSED', '1') != '0':
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                from .comm import GradientExchange
                from .utils import switches as _sw0
                ex = GradientExchange(info.grad_floats, 0 if _sw0.off('dp_fold_exchanges') else info.exchange_floats)
                if ex.probe():
                    self.exchange, self.grads = ex, ex.arena[:info.grad_floats]
                else:
                    if dist.get_rank() == 0:
                        warnings.warn('rlrep_amd: the in-launch gradient exchange did not pass its probe on this set of ranks '
                                      f'(fine-grained block: {ex.fine_grained}, ranks on one device: {ex.same_device}, {ex.error or "a peer reported the failure"}); gradients go through torch.distributed')
                    ex.close()
        self.exp_avg, self.exp_avg_sq = carve(3, torch.float32), carve(4, torch.float32)
        self.workspace = carve(5, torch.uint8)
        self.alpha_state = carve(6, torch.float64)     # log_alpha, m, v, step
        self.alpha_state[0] = float(np.log(0.1))
        ar = _lib.Arenas(self.params.data_ptr(), self.targets.data_ptr(), self.grads.data_ptr(),
                         self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.workspace.data_ptr(),
                         self.alpha_state.data_ptr())
        h = C.c_void_p()
        torch.cuda.synchronize()
        check(lib.rlrep_agent_create(C.byref(self.dims), C.byref(self.hyper), C.byref(ar), _stream(), C.byref(h)), 'agent_create')
        self.h = h
        if self.exchange is not None:
            # gradient slices up to 4 MB (RLREP_ENABLE=dp_fused_mb=N) are summed inside their optimizer launch; larger ones (diffsrsac's 198 MB nabla-mu
            # group at Humanoid dims: bandwidth-bound, RCCL's ring is the right shape) keep the all-reduce between backward and apply.  Slices of at least 512 KB (RLREP_ENABLE=dp_two_shot_kb=N; 0: never) take the two-shot form
            # inside the launch when world >= 3: (N - 1) S / N bytes per phase and GPU instead of (N - 1) S (csrc/dp_pull.h).
            from .utils import switches as _sw
            cap = int(float(_sw.opt('dp_fused_mb', '4')) * (1 << 20)) // 4
            two = int(float(_sw.opt('dp_two_shot_kb', '512')) * 1024) // 4
            t = _sw.opt('dp_timeout_s')
            if t is not None:
                self.exchange.set_timeout(float(t))
            self.fused_groups = frozenset(self.exchange.attach(self.h, cap, two))
        names = (C.c_char * 32 * METRIC_SLOTS)()
        lib.rlrep_metric_names(self.dims.alg, C.cast(names, C.c_void_p), METRIC_SLOTS)
        self.metric_names = [bytes(n).split(b'\0', 1)[0].decode() for n in names]
        mp = lib.rlrep_metrics_dev(self.h)
        self._metrics_ptr = mp
        self._keep = []

    def __del__(self):
        try:
            if getattr(self, 'h', None):
                torch.cuda.synchronize()
                lib.rlrep_agent_destroy(self.h)
                self.h = None
            if getattr(self, 'exchange', None) is not None:
                self.grads = None
                if not hasattr(self.exchange.group, 'members'):          # (a loopback group owns its members)
                    self.exchange.close()
                self.exchange = None
        except Exception:
            pass

    # ---- named tensors -----------------------------------------------------------------------
    def view(self, name, which='param'):
        arena, group, off, rows, cols = self.descs[name]
        base = {'param': self.params if arena == _lib.ARENA_PARAM else self.targets,
                'grad': self.grads, 'exp_avg': self.exp_avg, 'exp_avg_sq': self.exp_avg_sq}[which]
        if which != 'param' and arena != _lib.ARENA_PARAM:
            raise KeyError(f'{name} is not trainable')
        t = base[off:off + rows * cols]
        return t.view(rows, cols) if (cols > 1 or name.endswith('weight') or name.endswith('noise')) else t.view(rows)

    def group_slice(self, group, with_tail=False):
        o, n = self.layout.group_offset[group], self.layout.group_floats[group]
        return self.grads[o:o + n]

    def grad_tail(self):
        return self.grads[self.layout.param_floats:self.layout.param_floats + _lib.GRAD_TAIL]

    def load_state(self, sd):
        """sd: name -> array/tensor with the reference's state_dict keys (+ 'log_alpha', 'critic.noise')."""
        with torch.no_grad():
            for k, v in sd.items():
                if k == 'log_alpha':
                    self.alpha_state[0] = float(np.asarray(v))
                    continue
                if k not in self.descs:
                    continue
                dst = self.view(k)
                dst.copy_(torch.as_tensor(np.asarray(v), dtype=torch.float32).reshape(dst.shape))

    def state(self):
        if getattr(self, 'before_read', None) is not None:
            self.before_read()             # e.g. the agent's flush() of a pending deferred critic / actor pair
        out = {k: self.view(k).detach().cpu().clone() for k in self.order}
        out['log_alpha'] = self.alpha_state[0].detach().cpu().clone()
        self.chain_check()
        return out

    def exchange_check(self):
        """Raise if a wait of the in-launch gradient exchange has run out since the last check (a peer that never arrived: the affected
        step is invalid).  A read of mapped host memory: no device synchronisation, cheap enough for every flush()."""
        if self.exchange is not None:
            # NOT cleared: the error is sticky and the rank stays poisoned on the device (its exchange launches apply nothing) until the caller has
            # re-synchronised the replicas and calls core.exchange.status(clear=True) itself -- catching this exception must not resume training
            self.exchange.status(raise_on_error=True, clear=False)

    def chain_check(self):
        """Raise if a persistent chain launch (csrc/xchain.hip) failed its device-side checks since the agent was created: a wait that
        timed out, or workgroups of one group on different XCDs.  Synchronises the current stream."""
        if not _lib.has_experiments():          # only the experimental engines (row programs, chain kernel) ever set the word
            return
        w = C.c_uint32(0)
        check(lib.rlrep_chain_status(self.h, C.byref(w), _stream()), 'chain_status')

    # ---- data ---------------------------------------------------------------------------------
    def set_batch(self, slot, state, action, reward, next_state, done):
        B = int(state.shape[0])
        ts = [t.to(self.device, torch.float32).contiguous() for t in (state, action, reward, next_state, done)]
        self._keep = ts
        b = _lib.Batch(ts[0].data_ptr(), ts[1].data_ptr(), ts[2].data_ptr(), ts[3].data_ptr(), ts[4].data_ptr(), B)
        check(lib.rlrep_set_batch(self.h, slot, C.byref(b), _stream()), 'set_batch')

    def sample(self, slot, ring, idx, batch):
        check(lib.rlrep_replay_sample(self.h, slot, _ptr(ring), _ptr(idx), int(batch), _stream()), 'replay_sample')

    def prefetch_batch(self, ring, idx, batch, slot=0):
        """Arm the gather of the next minibatch of `slot` to ride in the next optimizer launch."""
        rc = lib.rlrep_prefetch_batch_slot(self.h, int(slot), _ptr(ring), _ptr(idx), int(batch))
        if rc < 0:
            check(rc, 'prefetch_batch')
        return rc == 1

    def feature_chain_next(self):
        """The next feature step follows this one directly on the minibatch just armed by prefetch_batch: chain the two
        (include/rlrep.h rlrep_feature_chain_next).  False: no such form for this agent."""
        rc = lib.rlrep_feature_chain_next(self.h)
        if rc < 0:
            check(rc, 'feature_chain_next')
        return rc == 1

    def train_prologue(self, ring, size_dev, idx_pool, eps_pool, seed, idx_offset, eps_offset, batch):
        """begin_train + index pool + noise pool + gather of the first minibatch, one launch."""
        check(lib.rlrep_train_prologue(self.h, _ptr(ring), _ptr(size_dev), _ptr(idx_pool), idx_pool.numel(), _ptr(eps_pool),
                                       eps_pool.numel(), int(seed), int(idx_offset), int(eps_offset), int(batch), _stream()),
              'train_prologue')

    # ---- steps --------------------------------------------------------------------------------
    def begin_train(self):
        check(lib.rlrep_begin_train(self.h, _stream()), 'begin_train')

    def feature_step(self, eps=None, noise_idx=None):
        check(lib.rlrep_feature_step(self.h, _ptr(eps), _ptr(noise_idx), _stream()), 'feature_step')

    def feature_backward(self, eps=None, noise_idx=None):
        check(lib.rlrep_feature_backward(self.h, _ptr(eps), _ptr(noise_idx), _stream()), 'feature_backward')

    def feature_exchange_count(self):
        return lib.rlrep_feature_exchange_count(self.h)

    def feature_exchange(self, k):
        """-> (kind, float32 tensor view of the library buffer, per-rank count, local offset)"""
        kind, ptr, count, off = C.c_int32(), C.c_void_p(), C.c_int64(), C.c_int64()
        check(lib.rlrep_feature_exchange(self.h, k, C.byref(kind), C.byref(ptr), C.byref(count), C.byref(off)), 'feature_exchange')
        if kind.value == 3:          # a finished slice of the gradient arena (local offset = its offset in the arena)
            return 3, self.grads[off.value:off.value + count.value], count.value, off.value
        total = count.value * (self.hyper.world_size if kind.value == 1 else 1)
        o = (ptr.value - self.workspace.data_ptr())
        buf = self.workspace[o:o + 4 * total].view(torch.float32)
        return kind.value, buf, count.value, off.value

    def feature_backward_part(self, part, eps=None, noise_idx=None):
        check(lib.rlrep_feature_backward_part(self.h, part, _ptr(eps), _ptr(noise_idx), _stream()), 'feature_backward_part')

    def feature_apply(self):
        check(lib.rlrep_feature_apply(self.h, _stream()), 'feature_apply')

    def prefetch_policy(self, eps_actor):
        """Arm the critic step to also run the forward half of the actor step (same batch, noise `eps_actor`)."""
        rc = lib.rlrep_prefetch_policy(self.h, _ptr(eps_actor))
        if rc < 0:
            check(rc, 'prefetch_policy')
        return rc == 1

    def prefetch_policy_early(self, eps_critic, eps_actor):
        """Arm the NEXT feature step (the last one of this train()) to also run both policy forwards."""
        rc = lib.rlrep_prefetch_policy_early(self.h, _ptr(eps_critic), _ptr(eps_actor))
        if rc < 0:
            check(rc, 'prefetch_policy_early')
        return rc == 1

    def critic_step(self, eps):
        check(lib.rlrep_critic_step(self.h, _ptr(eps), _stream()), 'critic_step')

    def critic_backward(self, eps):
        check(lib.rlrep_critic_backward(self.h, _ptr(eps), _stream()), 'critic_backward')

    def critic_apply(self):
        check(lib.rlrep_critic_apply(self.h, _stream()), 'critic_apply')

    def actor_step(self, eps):
        check(lib.rlrep_actor_alpha_step(self.h, _ptr(eps), _stream()), 'actor_alpha_step')

    def actor_backward(self, eps):
        check(lib.rlrep_actor_backward(self.h, _ptr(eps), _stream()), 'actor_backward')

    def actor_apply(self):
        check(lib.rlrep_actor_apply(self.h, _stream()), 'actor_apply')

    def update_target(self):
        check(lib.rlrep_update_target(self.h, _stream()), 'update_target')

    # ---- deferred critic / actor steps (include/rlrep.h: the feature steps of train(t+1) may run beside them) --------
    def defer_supported(self):
        """Number of snapshot sets the deferred critic / actor chain may rotate over (0: not built for this agent)."""
        return int(lib.rlrep_defer_supported(self.h))

    def defer_snapshot(self, eps_critic, eps_actor, set=0):
        check(lib.rlrep_defer_snapshot(self.h, int(set), _ptr(eps_critic), _ptr(eps_actor), _stream()), 'defer_snapshot')

    def defer_arm(self, eps_critic, eps_actor, set=0):
        """Before the LAST feature step of a train(): that step's optimizer launch also writes snapshot set `set` (rlrep_defer_arm);
        the defer_snapshot that follows with the same arguments launches nothing.  False: no folded form for this agent."""
        return lib.rlrep_defer_arm(self.h, int(set), _ptr(eps_critic), _ptr(eps_actor)) == 1

    def deferred_critic_actor(self, set=0):
        check(lib.rlrep_deferred_critic_actor(self.h, int(set), _stream()), 'deferred_critic_actor')

    def deferred_part(self, set, part):
        check(lib.rlrep_deferred_part(self.h, int(set), int(part), _stream()), 'deferred_part')

    def end_train(self):
        check(lib.rlrep_end_train(self.h), 'end_train')

    # ---- weight images of the vlsac noise critic (include/rlrep.h rlrep_images_managed) -----------------------------------------
    def images_managed(self, on):
        """While on, the step entry points do not launch the image refresh at the head of a critic step (captured train() graphs then
        carry one launch less; the caller refreshes after foreign writes).  True if this agent keeps images at all."""
        rc = lib.rlrep_images_managed(self.h, 1 if on else 0)
        if rc < 0:
            check(rc, 'images_managed')
        return rc == 1

    def refresh_images(self):
        check(lib.rlrep_refresh_images(self.h, _stream()), 'refresh_images')

    def arena_versions(self):
        """torch's in-place version counters of the parameter / target arenas: every torch write through ANY view of them (load_state_dict,
        `.data.copy_`, an optimizer of the caller's) bumps one; the library's own kernels, which write through raw pointers, do not."""
        return (self.params._version, self.targets._version)

    def sync_frozen(self):
        check(lib.rlrep_sync_frozen(self.h, _stream()), 'sync_frozen')

    def actor_forward(self, obs, eps, lo, hi, out=None):
        n = int(obs.shape[0])
        if out is None:
            out = torch.empty(n, self.dims.action_dim, dtype=torch.float32, device=self.device)
        check(lib.rlrep_actor_forward(self.h, _ptr(obs), n, _ptr(eps), float(lo), float(hi), _ptr(out), _stream()), 'actor_forward')
        return out

    def select_action(self, obs_pin, explore, seed, offset, lo, hi, act_pin):
        """One observation -> one action in one launch; both buffers pinned host tensors the kernel reads / writes in place (rlrep_select_action)."""
        check(lib.rlrep_select_action(self.h, _ptr(obs_pin), 1, 1 if explore else 0, int(seed), int(offset), float(lo), float(hi), _ptr(act_pin), 1, _stream()), 'select_action')

    # ---- noise --------------------------------------------------------------------------------
    def fill_normal(self, t, std, seed, offset):
        check(lib.rlrep_fill_normal(_ptr(t), t.numel(), float(std), int(seed), int(offset), _stream()), 'fill_normal')

    def fill_indices(self, t, hi, seed, offset):
        check(lib.rlrep_fill_indices(_ptr(t), t.numel(), int(hi), int(seed), int(offset), _stream()), 'fill_indices')

    def fill_normal_dev(self, t, std, seed, offset):
        check(lib.rlrep_fill_normal_dev(_ptr(t), t.numel(), float(std), int(seed), int(offset),
                                        C.c_void_p(lib.rlrep_steps_dev(self.h)), _stream()), 'fill_normal_dev')

    def fill_indices_dev(self, t, hi_dev, seed, offset):
        check(lib.rlrep_fill_indices_dev(_ptr(t), t.numel(), _ptr(hi_dev), int(seed), int(offset),
                                         C.c_void_p(lib.rlrep_steps_dev(self.h)), _stream()), 'fill_indices_dev')

    # ---- metrics ------------------------------------------------------------------------------
    def metrics_tensor(self):
        """A float32[16] torch view of the library's metric slots (no copy)."""
        if not hasattr(self, '_mt'):
            off = self._metrics_ptr - self.workspace.data_ptr()
            self._mt = self.workspace[off:off + 4 * METRIC_SLOTS].view(torch.float32)
        return self._mt

    # ---- metric history (include/rlrep.h rlrep_history): the metrics of whole-train() graph replays, read when somebody looks --------
    def history(self, on):
        check(lib.rlrep_history(self.h, 1 if on else 0), 'history')

    def _history_views(self):
        if not hasattr(self, '_hist'):
            ring, seq = C.c_void_p(), C.c_void_p()
            n, rec, tag = C.c_int32(), C.c_int32(), C.c_int32()
            check(lib.rlrep_history_dev(self.h, C.byref(ring), C.byref(seq), C.byref(n), C.byref(rec), C.byref(tag)), 'history_dev')
            base = self.workspace.data_ptr()
            r = self.workspace[ring.value - base:ring.value - base + 4 * n.value * rec.value].view(torch.float32).view(n.value, rec.value)
            q = self.workspace[seq.value - base:seq.value - base + 4].view(torch.int32)
            self._hist = (r, q, n.value, tag.value)
        return self._hist

    def history_seq(self):
        """Number of records filed so far (SYNCHRONISES the current stream)."""
        return int(self._history_views()[1].item())

    def history_source(self, n):
        """LazyInfo source for the metrics of the n-th filed train(): fetched from the ring on first read."""
        ring, _, cap, tag = self._history_views()

        def fetch(host_ring=None):
            rec = (host_ring if host_ring is not None else ring)[n % cap].cpu()
            got = int(rec.view(torch.int32)[tag])
            if got != n:
                raise RuntimeError(f'the metrics of this train() call (record {n}) have been overwritten: the history ring holds the last {cap} '
                                   f'calls (found record {got}); read a returned info dict within {cap} train() calls')
            return rec[:METRIC_SLOTS]
        fetch.history_record = n
        return fetch

    def history_capacity(self):
        return self._history_views()[2]

    def history_resolve(self):
        """Resolve every returned-but-unread info dict that is backed by the history ring: ONE copy of the ring to the host, then each live
        dict takes its record.  The agent calls this every capacity / 2 replays, so a dict a caller keeps (per-episode / per-epoch logging)
        stays valid forever, like the reference's plain floats -- at the price of one device synchronisation per 512 train() calls, and
        only while unread dicts are alive."""
        live = [li for li in (r() for r in getattr(self, '_hist_unread', ())) if li is not None and not li._done]
        self._hist_unread = []
        if not live:
            return
        host = self._history_views()[0].cpu()
        for li in live:
            li._fetch(host)

    def info(self, keys=None, lazy_source=None, early=None, on_read=None):
        snap = lazy_source if lazy_source is not None else self.metrics_tensor().clone()
        names = self.metric_names if keys is None else [n if n in keys else '' for n in self.metric_names]
        li = LazyInfo(names, snap, early, after=self.chain_check, on_read=on_read)
        if getattr(lazy_source, 'history_record', None) is not None:
            if not hasattr(self, '_hist_unread'):
                self._hist_unread = []
            self._hist_unread.append(weakref.ref(li))        # (a dict subclass is unhashable: no WeakSet)
        return li

    def stages(self, program):
        n = lib.rlrep_stage_count(self.h, program)
        return [lib.rlrep_stage_name(self.h, program, i).decode() for i in range(n)]

    def run_stage(self, program, stage):
        check(lib.rlrep_run_stage(self.h, program, stage, _stream()), 'run_stage')

    def device_state(self):
        """uint8 view of the library's batch-independent device state (train() counter, per-group Adam step
        counters and scalars, metric slots): what a checkpoint must carry besides the arenas."""
        end = self._metrics_ptr - self.workspace.data_ptr() + 4 * METRIC_SLOTS
        return self.workspace[:end]

    def sync_step_mirror(self):
        """After foreign writes into the device records (a checkpoint load): word 2 of the train() counter block -- what the next train prologue
        reads -- follows word 0 (csrc/elementwise.hip train_prologue_kernel).  Snapshots written before the split carry a zero there."""
        off = lib.rlrep_steps_dev(self.h) - self.workspace.data_ptr()
        w = self.workspace[off:off + 32].view(torch.int32)
        w[2] = w[0]

    def group_cfg(self):
        """float32[4, 22] view of the optimizer groups' device records (include/rlrep.h rlrep_group_cfg_dev): column 0 is the int32
        step counter (bit pattern), columns 1..5 = lr, beta1, beta2, eps, tau."""
        off = lib.rlrep_group_cfg_dev(self.h) - self.workspace.data_ptr()
        return self.workspace[off:off + 4 * 22 * 4].view(torch.float32).view(4, 22)

    def launch_count(self):
        return lib.rlrep_last_launch_count(self.h)
