"""SACAgent with the reference's call surface (agent/sac/sac_agent.py:15-188) on the HIP step programs.

`train(buffer, batch_size)`, `critic_step(batch)`, `update_actor_and_alpha(batch)`, `update_target()`,
`select_action(state, explore)` and the constructor kwargs are the reference's; the arithmetic runs in
librlrep_hip.so (rlrep_amd/csrc).  Subclasses (vlsac, ctrlsac, spedersac, diffsrsac) only describe their
dimensions, hyper-parameters, feature-step noise and parameter initialisation.
"""
import contextlib
import gc
import os
import zlib
import numpy as np
import torch
from torch import nn

from rlrep_amd.utils import switches as _sw
from rlrep_amd.core import HipCore
from rlrep_amd.utils import util
from rlrep_amd.utils.streams import raw_stream as _raw_stream, current_stream as _current_stream, on_stream as _on_stream

device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')


@contextlib.contextmanager
def _no_gc():
    """Around every hipGraph capture: collect now, then keep the cyclic collector off until the captures are done.  A collection that
    fires INSIDE a capture can finalise an agent of an earlier life (HipCore.__del__: device synchronise + hipFree) or a CUDAGraph --
    calls that are not permitted while a stream of the process is capturing; the process then dies in the runtime (seen as a bare
    `Aborted` with "Garbage-collecting" on top of the Python stack, depending on nothing but the allocation count)."""
    was = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


class ArenaModule(nn.Module):
    """nn.Module shell whose parameters are views into the agent's flat device arenas, registered under
    the reference's names, so `state_dict()` / `load_state_dict()` / `parameters()` keep working."""

    def __init__(self, core, prefix):
        super().__init__()
        self._core_ref = [core]
        for name in core.order:
            if not name.startswith(prefix + '.') or name.endswith('.noise'):
                continue
            parts = name[len(prefix) + 1:].split('.')
            mod = self
            for p in parts[:-1]:
                if not hasattr(mod, p):
                    mod.add_module(p, nn.Module())
                mod = getattr(mod, p)
            mod.register_parameter(parts[-1], nn.Parameter(core.view(name), requires_grad=False))

    def _quiesce(self):
        # the parameters are raw arena views: a pipelined train() may still be writing them on its own streams
        hook = getattr(self._core_ref[0], 'before_read', None)
        if hook is not None:
            hook()
            torch.cuda.current_stream().synchronize()

    def state_dict(self, *args, **kwargs):
        self._quiesce()
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self._quiesce()
        return super().load_state_dict(*args, **kwargs)


_STREAM_PAIR = {}


def _concurrent_stream_pair(core, index=0):
    """Two torch streams that really run side by side.  ROCm maps HIP streams onto a handful of hardware queues (4 by default; this package
    asks for 16) and two streams that share one are serialised -- measured here: the same two launch chains overlapped 1.8x or 1.00x
    depending only on which streams they were given.  So candidates are TIMED: two short kernel chains back to back on one stream against
    one chain on each stream of a pair; streams that overlap with every stream kept so far are kept for the life of the process.
    index > 0 (several replicas inside one process: the loopback form of rlrep_amd/comm.py): pair `index` of a set whose members are ALL
    mutually concurrent -- a replica's optimizer launch waits on the device for its peers', which must not be queued behind it."""
    dev = core.device
    kept = _STREAM_PAIR.setdefault(dev, [])
    need = 2 * (index + 1)
    if len(kept) >= need:
        return kept[2 * index], kept[2 * index + 1]
    import time
    bufs = [torch.empty(1 << 21, device=dev) for _ in range(2)]

    def chain(k):
        for _ in range(6):
            core.fill_normal(bufs[k], 1.0, 1, 2)

    def timed(fn):
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize(dev)
            best = min(best, time.perf_counter() - t0)
        return best

    first = kept[0] if kept else torch.cuda.Stream(device=dev)

    def serial():
        with torch.cuda.stream(first):
            chain(0); chain(1)

    serial()
    t_seq = timed(serial)
    if not kept:
        kept.append(first)
    tried = 0
    while len(kept) < need and tried < 24:
        cand = torch.cuda.Stream(device=dev)
        tried += 1
        ok = True
        for other in kept:
            def both(a=other, b=cand):
                with torch.cuda.stream(a):
                    chain(0)
                with torch.cuda.stream(b):
                    chain(1)
            if not timed(both) < 0.72 * t_seq:
                ok = False
                break
        if ok:
            kept.append(cand)
    while len(kept) < need:            # nothing (more) overlapped (single hardware queue?): still correct for ONE replica, just not concurrent
        if index > 0:
            raise RuntimeError('rlrep_amd: could not find %d mutually concurrent HIP streams for the loopback replicas (GPU_MAX_HW_QUEUES?)' % need)
        kept.append(torch.cuda.Stream(device=dev))
    return kept[2 * index], kept[2 * index + 1]


def _world():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(), dist.get_rank()
    except Exception:
        pass
    return 1, 0


class SACAgent(object):
    ALG = 'sac'
    PREFETCH_CHAIN = True     # every pooled index key is a slot-0 minibatch, consumed in _plan() order
    MODULES = ('critic', 'critic_target', 'actor')
    FEATURE_KEYS = ()
    CRITIC_KEYS = ('q_loss', 'q1', 'q2')
    ACTOR_KEYS = ('actor_loss', 'alpha_loss', 'alpha')
    CHECKPOINT_FORMAT = 'rlrep-ckpt-2'        # 2: GroupCfg grew to 22 words (running Adam powers); bump with the device-record layout

    def __init__(self, state_dim, action_dim, action_space, lr=3e-4, discount=0.99, target_update_period=2,
                 tau=0.005, alpha=0.1, auto_entropy_tuning=True, hidden_dim=1024, **_hip):
        self._init_common(state_dim, action_dim, action_space, discount, target_update_period, tau, alpha,
                          auto_entropy_tuning)
        self._dims = dict(state_dim=state_dim, action_dim=action_dim, hidden_dim=hidden_dim,
                          actor_hidden_dim=hidden_dim)
        self._hyper = dict(lr_feature=lr, lr_critic=lr, lr_actor=lr)
        self._finish_init(_hip)

    # ---- construction -------------------------------------------------------------------------
    def _init_common(self, state_dim, action_dim, action_space, discount, target_update_period, tau, alpha,
                     auto_entropy_tuning):
        self.steps = 0
        self.device = device
        self.state_dim, self.action_dim = int(state_dim), int(action_dim)
        self.action_range = [float(action_space.low.min()), float(action_space.high.max())]
        self.discount, self.tau = float(discount), float(tau)
        self.target_update_period = int(target_update_period)
        self.learnable_temperature = bool(auto_entropy_tuning)
        self.target_entropy = -action_dim
        self._alpha0 = float(alpha)
        self.extra_feature_steps = 0
        self.feature_tau = 0.0

    def _finish_init(self, hip_kwargs):
        self.max_batch = int(hip_kwargs.get('max_batch', os.environ.get('RLREP_MAX_BATCH', 256)))
        self.world_size, self.rank = _world()
        # loopback=(LoopbackGroup, rank): this agent is one of several replicas inside ONE process (rlrep_amd/comm.py; tools/exp/dp_loopback.py, tests)
        self._loopback = hip_kwargs.get('loopback')
        if self._loopback is not None:
            self.world_size, self.rank = self._loopback[0].world, int(self._loopback[1])
        dims = dict(feature_dim=0, vae_hidden_dim=0, phi_hidden_dim=0, phi_hidden_depth=0, mu_hidden_dim=0,
                    mu_hidden_depth=0, num_noise=0, max_batch=self.max_batch, flags=0)
        dims.update(self._dims)
        dims['rank'] = self.rank
        hyper = dict(discount=self.discount, tau=self.tau, feature_tau=float(self.feature_tau),
                     target_entropy=float(self.target_entropy), sigma_scale=0.0,
                     target_update_period=self.target_update_period,
                     extra_feature_steps=int(self.extra_feature_steps), learn_alpha=int(self.learnable_temperature))
        hyper.update({k: float(v) for k, v in self._hyper.items()})
        self.core = HipCore(self.ALG, dims, hyper, world_size=self.world_size, loopback=self._loopback)
        self.core.alpha_state[0] = float(np.log(self._alpha0))      # quirk Q1: float64 log_alpha
        self._init_parameters()
        if self._loopback is not None:
            grp = self._loopback[0]
            if self.rank == 0:
                grp.reference_core = self.core
            else:
                for name in ('params', 'targets', 'alpha_state'):
                    getattr(self.core, name).copy_(getattr(grp.reference_core, name))
        elif self.world_size > 1:
            import torch.distributed as dist
            for t in (self.core.params, self.core.targets, self.core.alpha_state):
                dist.broadcast(t, src=0)
        for m in self.MODULES:
            setattr(self, m, ArenaModule(self.core, m))
        self._seed = int(hip_kwargs.get('seed', torch.initial_seed() & 0x7fffffff)) + 7919 * self.rank
        self._ctr = 0
        self._bufs = {}
        self._graph = None
        self._seg = None
        self._seg_capture_colls = False
        self._early_works, self._early_slices = [], []       # async gradient all-reduces issued inside a backward (exchange kind 3)
        self._n_captured_colls = 0
        self._fresh_dp_graph = False
        self.use_graph_dp = not _sw.off('graph_dp')
        # sequential data-parallel form on the RCCL backend: the gradient all-reduces may be CAPTURED into the hipGraph of train() (one graph
        # instead of 7 segments + 6 eager calls of ~14 us of host time each; one stream, one communicator: the collectives of a replay
        # are issued in program order, identical on every rank).  That form has only ever run over a ONE-rank RCCL group (the only RCCL
        # configuration a one-GPU box allows), where an all-reduce moves nothing between devices -- so it is the default only there.
        # With world_size > 1 the default is the form that two real ranks have run (gloo, tests/test_dp.py): graph segments around
        # eager all-reduces.  RLREP_DP_CAPTURE=1 (bench.py --dp-form captured) opts in; a failure while building or first replaying the
        # captured graph raises with that switch named (no silent fallback: every rank must take the same form).
        self.capture_collectives = bool(int(os.environ.get('RLREP_DP_CAPTURE', '1' if self.world_size == 1 else '0')))
        # backward -> all-reduce -> apply form of every optimizer step.  RLREP_FORCE_DP=1 takes it with a one-rank process group too
        # (sac / vlsac: rehearses the RCCL stream / graph-segment machinery on a single GPU; tests/test_dp.py)
        self._dp = self.world_size > 1 or (self.ALG in ('sac', 'vlsac', 'diffsrsac') and bool(int(os.environ.get('RLREP_FORCE_DP', '0'))))
        # every gradient slice of this agent is summed inside its optimizer launch (HipCore.fused_groups) and its backward needs no collective of
        # its own: a data-parallel train() is then the SAME sequence of launches as on one GPU, and takes the single-GPU graph forms
        # (one graph, or the two chains on two streams) -- nothing for the host to do between two optimizer steps
        lay = self.core.layout
        self._fused_all = (self.world_size > 1 and all(self._fused(g) for g in range(4) if lay.group_floats[g] > 0)
                           and self.core.feature_exchange_count() == 0)
        self._inject = None
        self._prepare_only = False
        self._pool = None
        self._next_key = {}
        self._early_key = None
        self.use_graph = bool(int(os.environ.get('RLREP_GRAPH', '1'))) and hip_kwargs.get('graph', True)
        # critic / actor steps of train(t) as a graph branch beside the feature steps of train(t+1) (vlsac; _train_graph_pipelined)
        self.use_pipeline = bool(int(os.environ.get('RLREP_PIPELINE', '1'))) and hip_kwargs.get('pipeline', True)
        # The two-chain schedule pays when train() calls follow each other; a caller that LOOKS at the critic / actor between two calls -- main.py's
        # loop: select_action before every train() -- ends the overlap each time, and what is left of the schedule is its cost (snapshot, event
        # hand-off between the streams: 466 us per call against 431 us for the one-graph sequential train(); tools/exp/rl_loop.py: 1 600 vs 1 880
        # iterations/s).  Both forms perform identical updates (tests), so train() picks per call: three calls in a row whose pair was waited
        # for before the next call -> the sequential graph; two calls in a row with nothing looked at in between -> back to the two chains.
        self._pipeline_mode = int(os.environ.get('RLREP_PIPELINE', '2'))          # (read once: the per-call paths do not touch os.environ)
        self._stamp_on = _sw.opt('stamp') is not None
        self._adaptive = not _sw.off('adaptive_pipeline') and hip_kwargs.get('adaptive', 'pipeline' not in hip_kwargs)      # (an explicit pipeline= argument pins the form)
        self._looked, self._n_looked, self._n_b2b = False, 0, 0
        # two communicators in flight (one per chain): never met a second real rank on hardware, so it is opt-in (RLREP_PIPELINE_DP=1);
        # the default N > 1 form is the sequential one (one communicator, program order identical on every rank)
        self.use_pipeline_dp = bool(int(os.environ.get('RLREP_PIPELINE_DP', '0')))
        self._pg_ca = None           # second process group (own communicator / stream) for the deferred chain's all-reduces; see _train_graph_dp_pipelined
        self._pipe = None
        self._pending = False
        self.core.before_read = self.flush
        # weight images of the noise critic (vlsac): captured single-GPU train() graphs leave their upkeep to the optimizer launches and to
        # _sync_images() (one launch less per train(): DESIGN.md 5.5)
        self._img_on = False               # some captured graph relies on managed images
        self._img_dirty = True
        self._img_seen = None

    # parameter initialisation (values only; layout is the library's)
    def _orth(self, name, gain=1.0):
        w = self.core.view(name)
        cpu = torch.empty(w.shape)
        nn.init.orthogonal_(cpu, gain)
        w.copy_(cpu)

    def _default_linear(self, wname, bname):
        """nn.Linear default init: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and bias."""
        w = self.core.view(wname)
        bound = 1.0 / np.sqrt(w.shape[1])
        w.copy_(torch.empty(w.shape).uniform_(-bound, bound))
        b = self.core.view(bname)
        b.copy_(torch.empty(b.shape).uniform_(-bound, bound))

    def _init_prefix(self, prefix, orthogonal):
        for n in self.core.order:
            if n.startswith(prefix + '.') and n.endswith('.weight'):
                if orthogonal:
                    self._orth(n)
                    self.core.view(n[:-6] + 'bias').zero_()
                else:
                    self._default_linear(n, n[:-6] + 'bias')

    def _copy_prefix(self, src, dst):
        for n in self.core.order:
            if n.startswith(src + '.') and not n.endswith('noise'):
                self.core.view(dst + n[len(src):]).copy_(self.core.view(n))

    def _init_parameters(self):
        # utils/util.py:61-66 via .apply(weight_init) in DoubleQCritic / DiagGaussianActor
        self._init_prefix('critic', True)
        self._copy_prefix('critic', 'critic_target')            # sac_agent.py:58
        self._init_prefix('actor', True)

    # ---- reference surface --------------------------------------------------------------------
    @property
    def alpha(self):
        return self.core.alpha_state[0].exp()

    @property
    def log_alpha(self):
        return self.core.alpha_state[0]

    def select_action(self, state, explore=False):
        return self._select_action(state, explore)

    def _select_action(self, state, explore=False, eps=None):
        """sac_agent.py:89-96.  `eps` (tests): the [1, A] standard-normal draw of `dist.sample()` instead of a fresh one."""
        self.flush()
        # one observation in, one action out, once per environment step: pinned host buffers and fixed device buffers (a pageable tensor costs a
        # staging copy and a synchronisation in each direction, and an allocation: tools/exp/rl_loop.py)
        sel = getattr(self, '_sel', None)
        if sel is None:
            dev = self.core.device
            sel = self._sel = dict(obs_pin=torch.empty(1, self.state_dim, dtype=torch.float32).pin_memory(),
                                   act_pin=torch.empty(1, self.action_dim, dtype=torch.float32).pin_memory(),
                                   obs=torch.empty(1, self.state_dim, dtype=torch.float32, device=dev),
                                   out=torch.empty(1, self.action_dim, dtype=torch.float32, device=dev))
        sel['obs_pin'].numpy()[0, :] = np.asarray(state, dtype=np.float32).reshape(-1)
        if eps is None:
            # ONE launch (rlrep_select_action): the kernel reads the pinned observation and writes the pinned action in place; the draw is the
            # one _noise('sel', ...) would have produced (same seed, same counter)
            if explore:
                self._ctr += 1
            self.core.select_action(sel['obs_pin'], explore, self._seed, self._ctr << 20, *self.action_range, sel['act_pin'])
            _current_stream().synchronize()
            return sel['act_pin'].numpy()[0].copy()
        sel['obs'].copy_(sel['obs_pin'], non_blocking=True)
        if explore and eps is not None:
            eps = torch.as_tensor(np.asarray(eps, dtype=np.float32)).reshape(1, self.action_dim).to(self.core.device)
        elif explore:
            eps = self._noise('sel', (1, self.action_dim))
        else:
            eps = None
        self.core.actor_forward(sel['obs'], eps, *self.action_range, out=sel['out'])
        sel['act_pin'].copy_(sel['out'], non_blocking=True)
        _current_stream().synchronize()
        return sel['act_pin'].numpy()[0].copy()

    def update_target(self):
        self.flush()
        self._img_dirty = True
        self.core.update_target()

    def critic_step(self, batch, eps=None):
        self.flush()
        self._img_dirty = True
        self._set_batch(batch)
        self.core.critic_step(self._noise('crit', (self._B, self.action_dim)) if eps is None else eps)
        return self.core.info(self.CRITIC_KEYS)

    def update_actor_and_alpha(self, batch, eps=None):
        self.flush()
        self._set_batch(batch)
        self.core.actor_step(self._noise('act', (self._B, self.action_dim)) if eps is None else eps)
        return self.core.info(self.ACTOR_KEYS)

    def train(self, buffer, batch_size):
        """One train step (sac_agent.py:169-188).  In pipelined graph mode (vlsac / ctrlsac / spedersac on one GPU) the critic and
        actor steps of this call may still be in flight when it returns; everything that looks at them waits (`flush()`), including
        reading the returned info dict -- which therefore has to be read before the NEXT train() call to describe THIS one."""
        self.steps += 1
        if (self.steps & 31) == 0:
            self.core.exchange_check()           # (a word in mapped host memory: no synchronisation; a launch that saw a timeout applied nothing)
        if self.use_graph and (not self._dp or self._fused_all):
            if self.use_pipeline and self._feature_iters() > 0 and self.core.defer_supported():
                if self._prefer_sequential():
                    self.flush()
                    out = self._train_graph(buffer, batch_size)
                else:
                    out = self._train_graph_pipelined(buffer, batch_size)
                self._looked = False             # (flush() calls made by the call itself do not count)
                return out
            return self._train_graph(buffer, batch_size)
        if self.use_graph and self.use_graph_dp:
            if self.use_pipeline and self.use_pipeline_dp and self._feature_iters() > 0 and self.ALG == 'vlsac' and self.core.defer_supported():
                return self._train_graph_dp_pipelined(buffer, batch_size)
            return self._train_graph_dp(buffer, batch_size)
        return self._train_eager(buffer, batch_size)

    update = train      # BASELINE.json's north_star calls it agent.update()

    def prepare(self, buffer, batch_size):
        """Capture the hipGraphs the next train(buffer, batch_size) will replay, WITHOUT launching anything of a train().  Several replicas in
        one process (the loopback form: rlrep_amd/comm.py LoopbackGroup) must build their graphs one after the other -- a capture synchronises
        the device, which would wait for a peer's optimizer launch that in turn waits for this replica's -- and replay them side by side
        afterwards.  No-op for the forms that are not whole graphs."""
        if not (self.use_graph and (not self._dp or self._fused_all)):
            return
        self._prepare_only = True
        try:
            if self.use_pipeline and self._feature_iters() > 0 and self.core.defer_supported():
                self._train_graph_pipelined(buffer, batch_size)         # (the two-chain form ...
            self._train_graph(buffer, batch_size)                       #  ... and the one-graph form train() switches to for a caller that looks at the actor between calls)
        finally:
            self._prepare_only = False

    # ---- checkpoint / resume (absent in the reference: `--save_model` is parsed and never read, main.py:37) ------
    def state_snapshot(self):
        c = self.core
        self.flush()
        torch.cuda.synchronize()
        c.exchange_check()                       # behind the device synchronisation: a replica whose last launches timed out is never written out
        return {'format': self.CHECKPOINT_FORMAT, 'device_state_bytes': int(c.device_state().numel()), 'alg': self.ALG, 'params': c.params.cpu(), 'targets': c.targets.cpu(), 'exp_avg': c.exp_avg.cpu(),
                'exp_avg_sq': c.exp_avg_sq.cpu(), 'alpha_state': c.alpha_state.cpu(), 'device_state': c.device_state().cpu(),
                'steps': self.steps, 'noise_ctr': self._ctr, 'seed': self._seed, 'layout': list(c.order)}

    def save(self, path):
        snap = self.state_snapshot()
        self.core.exchange_check()               # (the copies above were synchronous; refuse to persist a replica that is out of step)
        torch.save(snap, path)

    def load(self, path_or_snapshot):
        snap = torch.load(path_or_snapshot) if isinstance(path_or_snapshot, (str, bytes, os.PathLike)) else path_or_snapshot
        self.flush()                                   # nothing of a pipelined train() may still be writing the arenas
        torch.cuda.synchronize()
        c = self.core
        if snap['alg'] != self.ALG or snap['layout'] != list(c.order) or snap['params'].numel() != c.params.numel():
            raise RuntimeError('checkpoint does not match this agent (algorithm / dimensions differ)')
        # the device records (step counters, per-group optimizer scalars, metric slots) are copied back byte for byte: their layout is the
        # library's (include/rlrep.h RLREP_GROUP_CFG_WORDS) and changes with it -- a checkpoint of another format is refused by name, not
        # by an opaque copy_ shape error
        fmt, nbytes = snap.get('format'), snap.get('device_state_bytes', snap['device_state'].numel())
        if fmt is None and int(nbytes) == int(c.device_state().numel()):
            fmt = self.CHECKPOINT_FORMAT       # a snapshot written before the key existed, with device records of exactly this layout's size
        if fmt != self.CHECKPOINT_FORMAT or int(nbytes) != int(c.device_state().numel()) or snap['device_state'].numel() != c.device_state().numel():
            raise RuntimeError(f'checkpoint does not match this library: format {fmt!r} with {int(nbytes)} bytes of device records, this build '
                               f'writes format {self.CHECKPOINT_FORMAT!r} with {int(c.device_state().numel())} (saved by another version of rlrep_amd)')
        for k, dst in (('params', c.params), ('targets', c.targets), ('exp_avg', c.exp_avg), ('exp_avg_sq', c.exp_avg_sq),
                       ('alpha_state', c.alpha_state)):
            dst.copy_(snap[k])
        hyper = c.group_cfg()[:, 1:6].clone()          # lr, betas, eps, tau are THIS agent's constructor arguments, not the checkpoint's:
        c.device_state().copy_(snap['device_state'])   # the device records come back with the checkpoint's step counters only
        c.group_cfg()[:, 1:6].copy_(hyper)
        c.sync_step_mirror()
        self.steps, self._ctr, self._seed = snap['steps'], snap['noise_ctr'], snap['seed']
        self._img_dirty = True
        self._graph = None
        self._pipe, self._pending = None, False
        torch.cuda.synchronize()

    # ---- internals ----------------------------------------------------------------------------
    def _set_batch(self, batch, slot=0):
        self._B = int(batch.state.shape[0])
        self.core.set_batch(slot, batch.state, batch.action, batch.reward, batch.next_state, batch.done)

    def _buf(self, key, shape, dtype=torch.float32):
        t = self._bufs.get(key)
        if t is None or tuple(t.shape) != tuple(shape):
            t = self._bufs[key] = torch.empty(shape, dtype=dtype, device=self.core.device)
        return t

    def _noise(self, key, shape, std=1.0):
        t = self._buf('eps_' + key, shape)
        self._ctr += 1
        self.core.fill_normal(t, std, self._seed, self._ctr << 20)
        return t

    def _indices(self, key, n, hi):
        t = self._buf('idx_' + key, (n,), torch.int32)
        self._ctr += 1
        self.core.fill_indices(t, hi, self._seed, self._ctr << 20)
        return t

    def _feature_iters(self):
        return 0

    def _stream_pair(self):
        """The two streams of the two-chain train().  Several replicas inside one process (loopback) take disjoint pairs of ONE set of mutually
        concurrent streams, found by trial (rlrep_amd/comm.py concurrent_streams): a replica's optimizer launch waits on the device for its
        peers', which must not be queued behind it."""
        if getattr(self, '_stream_pair_override', None) is not None:
            return self._stream_pair_override
        if self._loopback is not None:
            from rlrep_amd.comm import concurrent_streams
            ss = concurrent_streams(2 * self.world_size)
            return ss[2 * self.rank], ss[2 * self.rank + 1]
        return _concurrent_stream_pair(self.core)

    def _feature_once(self, buffer, B, i, g):
        raise NotImplementedError

    def _fused(self, group):
        """The optimizer launch of `group` sums the ranks' gradients itself (rlrep_amd/comm.py, csrc/dp_pull.h): no collective to issue."""
        return group in getattr(self.core, 'fused_groups', ())

    def _allreduce(self, group, with_tail=False, pg=None):
        import torch.distributed as dist
        if self._fused(group):
            return
        lay = self.core.layout
        o, n = lay.group_offset[group], lay.group_floats[group]
        end = lay.grad_floats if with_tail else o + n
        view = self.core.grads[o:end]
        self._collective(lambda: dist.all_reduce(view, group=pg))

    def _collective(self, fn):
        """Run a torch.distributed call now, or -- while a data-parallel train() is being captured -- either record it into the open
        hipGraph (RCCL backend with RLREP_DP_CAPTURE=1: ProcessGroupNCCL under stream capture) or close the graph segment recorded so
        far, remember the collective as an eager step between two segments and open the next segment (the default with world_size > 1,
        and the only form on gloo)."""
        if self._seg is not None and self._seg_capture_colls:
            fn()                                    # recorded into the open graph (ProcessGroupNCCL under stream capture)
            self._n_captured_colls += 1
            return
        if self._seg is not None:
            segs, cur = self._seg
            cur.capture_end()
            segs.append(('graph', cur))
            segs.append(('coll', fn))
            nxt = torch.cuda.CUDAGraph()
            nxt.capture_begin(capture_error_mode='thread_local')
            self._seg = (segs, nxt)
            return
        fn()

    def _feature_backward_dp(self, eps=None, noise_idx=None):
        """Feature backward with the collectives that batch-coupled losses need inside it (ctrlsac: all-gather of
        mu(s') and all-reduce of its gradient; spedersac: all-reduce of Phibar and v); plain backward otherwise."""
        import torch.distributed as dist
        c = self.core
        n = c.feature_exchange_count()
        for k in range(n + 1):
            c.feature_backward_part(k, eps, noise_idx)
            if k < n:
                kind, buf, count, off = c.feature_exchange(k)
                if kind == 1:
                    views = [buf[r * count:(r + 1) * count] for r in range(self.world_size)]
                    local = views[self.rank]
                    self._collective(lambda views=views, local=local: dist.all_gather(views, local))
                elif kind == 3 and self._fused(self._group_of(off)):
                    pass                                 # its group's optimizer launch reads every rank's arena itself
                elif kind == 3:
                    # a slice of the gradient arena that is already final (diffsrsac: the nabla-mu head, 99 % of the bytes): its all-reduce is
                    # ISSUED here and travels under the rest of the backward; _wait_early_reduces() joins it before the optimizer launch
                    self._early_slices.append((off, off + count))
                    self._collective(lambda buf=buf: self._early_works.append(dist.all_reduce(buf, async_op=True)))
                else:
                    self._collective(lambda buf=buf: dist.all_reduce(buf))

    def _group_of(self, grad_offset):
        lay = self.core.layout
        for g in range(4):
            if lay.group_floats[g] > 0 and lay.group_offset[g] <= grad_offset < lay.group_offset[g] + lay.group_floats[g]:
                return g
        return -1

    def _wait_early_reduces(self):
        def join():
            for w in self._early_works:
                w.wait()
            self._early_works.clear()
        self._collective(join)

    def _allreduce_rest(self, group):
        """All-reduce what the early (kind 3) exchanges of this backward left of `group`'s gradient slice, then join them."""
        import torch.distributed as dist
        if self._fused(group):
            return
        lay = self.core.layout
        lo, hi = lay.group_offset[group], lay.group_offset[group] + lay.group_floats[group]
        done = sorted(s for s in self._early_slices if s[0] >= lo and s[1] <= hi)
        self._early_slices = [s for s in self._early_slices if not (s[0] >= lo and s[1] <= hi)]
        cur = lo
        for a, b in done + [(hi, hi)]:
            if a > cur:
                view = self.core.grads[cur:a]
                self._collective(lambda view=view: dist.all_reduce(view))
            cur = max(cur, b)
        self._wait_early_reduces()

    # ---- pooled noise: ALL sample indices and ALL standard-normal noise of one train() come from two
    # Philox launches into two contiguous buffers (instead of one launch per tensor) ---------------------
    def _plan(self, B):
        """(index draws, normal draws) of one train(): lists of keys / (key, shape), in consumption order."""
        return ['s0'], [('crit', (B, self.action_dim)), ('act', (B, self.action_dim))]

    def _fill_pools(self, buffer, B, g):
        idx_keys, eps_specs = self._plan(B)
        ni = len(idx_keys) * B
        ne = sum(int(np.prod(sh)) for _, sh in eps_specs)
        ipool = self._buf('pool_idx', (ni,), torch.int32)
        epool = self._buf('pool_eps', (ne,))
        if g and _sw.off('prologue'):
            self.core.fill_indices_dev(ipool, buffer.size_dev(), self._seed, 1 << 40)
            self.core.fill_normal_dev(epool, 1.0, self._seed, 2 << 40)
        elif g:
            # one launch: steps += 1, both pools and the gather of the first minibatch (first B pool indices)
            self.core.train_prologue(buffer.ring, buffer.size_dev(), ipool, epool, self._seed, 1 << 40, 2 << 40, B)
        else:
            self._ctr += 1
            self.core.fill_indices(ipool, buffer.size, self._seed, (1 << 40) + self._ctr)
            self.core.fill_normal(epool, 1.0, self._seed, (2 << 40) + self._ctr)
        self._pool = {}
        for q, k in enumerate(idx_keys):
            self._pool['idx_' + k] = ipool[q * B:(q + 1) * B]
        # slot-0 batches in the order they are consumed: after gathering one, the next is armed to ride in the
        # step's optimizer launch (rlrep_prefetch_batch)
        self._next_key = self._prefetch_chain(idx_keys) if g else {}
        o = 0
        for k, sh in eps_specs:
            n = int(np.prod(sh))
            self._pool['eps_' + k] = epool[o:o + n].view(*sh)
            o += n
        # the critic / actor steps reuse the last feature minibatch: both policy forwards can ride in its feature step
        self._early_key = idx_keys[-1] if (g and self.PREFETCH_CHAIN and self._feature_iters() > 0
                                           and 'eps_crit' in self._pool and 'eps_act' in self._pool) else None

    def _prefetch_chain(self, idx_keys):
        """key -> the key whose gather (same slot) rides in the optimizer launch of the step that consumes `key`."""
        return dict(zip(idx_keys, idx_keys[1:])) if self.PREFETCH_CHAIN else {}

    def _sample_into(self, buffer, B, key, slot=0, g=False):
        if self._inject is None and self._pool is not None and ('idx_' + key) in self._pool:
            self.core.sample(slot, buffer.ring, self._pool['idx_' + key], B)
            nxt = self._next_key.get(key)
            if nxt is not None:
                armed = self.core.prefetch_batch(buffer.ring, self._pool['idx_' + nxt], B, slot)
                # the feature step that consumes `nxt` follows this one directly: chain them (one launch less per pair; the library
                # declines where it has no such form) -- unless that next step is the one that carries the policy forwards
                if armed and slot == 0 and not self._dp and self._feature_iters() > 0 and nxt != self._early_key_for(key):
                    self.core.feature_chain_next()
            if slot == 0 and key == self._early_key:
                # this is the minibatch the critic and actor steps will reuse: both policy forwards ride in its feature step
                self.core.prefetch_policy_early(self._pool['eps_crit'], self._pool['eps_act'])
            return
        if self._inject is not None:
            idx = torch.as_tensor(np.asarray(self._inject['idx'].pop(0)), dtype=torch.int32).to(self.core.device)
            self._bufs['idx_' + key] = idx
        elif g:
            idx = self._buf('idx_' + key, (B,), torch.int32)
            self.core.fill_indices_dev(idx, buffer.size_dev(), self._seed, self._graph_off(key))
        else:
            idx = self._indices(key, B, buffer.size)
        self.core.sample(slot, buffer.ring, idx, B)

    def _early_key_for(self, key):
        """The key of the minibatch whose feature step will carry both policy forwards (rlrep_prefetch_policy_early), as far as it is
        known while `key` is being sampled: _fill_pools() has fixed it for this train()."""
        return self._early_key

    def _graph_off(self, key):
        # distinct Philox streams per noise tensor inside one train(): offset = (hash << 32) + device step counter
        return (zlib.crc32(key.encode()) % 65521 + 1) << 32

    def _eps(self, key, shape, g=False, std=1.0):
        if self._inject is None and self._pool is not None and ('eps_' + key) in self._pool and std == 1.0:
            return self._pool['eps_' + key]
        if self._inject is not None:
            e = self._inject['eps'].pop(0)
            t = torch.as_tensor(np.asarray(e)).to(self.core.device)
            t = t.to(torch.int32) if t.dtype in (torch.int64, torch.int32) else t.to(torch.float32)
            assert tuple(t.shape) == tuple(shape), (key, t.shape, shape)
            self._bufs['eps_' + key] = t.contiguous()
            return self._bufs['eps_' + key]
        if g:
            t = self._buf('eps_' + key, shape)
            self.core.fill_normal_dev(t, std, self._seed, self._graph_off(key))
            return t
        return self._noise(key, shape, std)

    def _body(self, buffer, B, g):
        """The whole train() as a sequence of stream-ordered library calls (captured into a hipGraph when g)."""
        c, W = self.core, (2 if self._dp else 1)
        self._pool = None
        self._next_key = {}
        self._early_key = None
        if self._inject is None and g and not _sw.off('prologue'):
            self._fill_pools(buffer, B, g)          # includes begin_train (rlrep_train_prologue)
        else:
            c.begin_train()
            if self._inject is None:
                self._fill_pools(buffer, B, g)
        nf = self._feature_iters()
        for i in range(nf):
            self._feature_once(buffer, B, i, g)
        if nf == 0:
            self._sample_into(buffer, B, 's0', 0, g)
        self._between_feature_and_critic()
        e1 = self._eps('crit', (B, self.action_dim), g)
        e2 = self._eps('act', (B, self.action_dim), g)
        c.prefetch_policy(e2)      # the actor step's forward half rides in the critic step's launches (same batch)
        if W > 1:
            if self._critic_trains():
                c.critic_backward(e1); self._allreduce(1); c.critic_apply()
            else:
                c.critic_step(e1)
        else:
            c.critic_step(e1)
        if W > 1:
            c.actor_backward(e2); self._allreduce(2, True); c.actor_apply()
        else:
            c.actor_step(e2)
        c.update_target()

    def _critic_trains(self):
        return True

    @contextlib.contextmanager
    def _managed_images(self):
        """Around the capture of a single-GPU train() graph: the critic steps recorded inside do not carry the image-refresh launch."""
        # Default since round 5 (RLREP_DISABLE=managed_images keeps the refresh launch at the head of every critic step).  Round 4 had measured it
        # at +0.3 % in the sequential form and -2.9 % in the two-chain form, when the feature chain alone bounded the period; with both chains
        # co-critical (docs/history/r05.md: a delay launch on either chain lengthens the period) the launch it takes off the critic / actor chain
        # returns +0.8 % (3 877 / 3 886 / 3 874 -> 3 914 / 3 916 / 3 899 train()/s, three alternations on one box).
        on = not _sw.off('managed_images') and self.core.images_managed(True)
        self._img_on = self._img_on or on
        try:
            yield
        finally:
            self.core.images_managed(False)

    def _sync_images(self):
        """Before replaying graphs that were captured with managed images: if anything but such a graph may have written critic /
        critic_target since the images were last known to be current -- an eager step method, a checkpoint load, any torch write into the
        arenas (version counters) -- regenerate them now (one eager launch; never in the steady state of a train() loop)."""
        if not self._img_on:
            return
        v = self.core.arena_versions()
        if self._img_dirty or v != self._img_seen:
            self.flush()
            self.core.refresh_images()
            self._img_dirty, self._img_seen = False, v

    def _history_info(self):
        """The info dict of a whole-train() graph replay: record n of the library's metric history ring, fetched when read.  Reading it
        counts as LOOKING at the critic / actor for the adaptive choice of the train() form (a caller who reads every dict and never calls
        select_action would otherwise oscillate between the forms: in the two-chain form a read flushes and marks the look, here it did
        not).  Before the ring can wrap, the unread dicts still alive are resolved from one host copy of it."""
        n = self._hist_n
        self._hist_n += 1
        half = max(1, self.core.history_capacity() // 2)
        if n % half == half - 1:
            self.core.history_resolve()
        return self.core.info(lazy_source=self.core.history_source(n), on_read=self._mark_looked)

    def _mark_looked(self):
        self._looked = True

    def _raise_capture_hint(self, e):
        """A data-parallel train() failed while its hipGraph was built or first replayed.  With the collectives CAPTURED into the graph
        (RLREP_DP_CAPTURE=1 on the RCCL backend) say how to get the segmented form instead -- every rank has to make the same choice, so
        there is no per-rank fallback, and a process that has touched the GPU is never re-executed: the launcher sets the switch."""
        if self._seg_capture_colls:
            raise RuntimeError('data-parallel train(): building / replaying the hipGraph with CAPTURED RCCL all-reduces failed '
                               f'({type(e).__name__}: {e}).  Re-launch with RLREP_DP_CAPTURE=0 (bench.py --dp-form segments): hipGraph '
                               'segments around eager all-reduces, the form the multi-rank tests run.') from e
        raise e

    def _between_feature_and_critic(self):
        pass

    def train_injected(self, buffer, batch_size, idx, eps):
        """train() with caller-supplied sample indices and noise tensors, consumed in the reference's draw
        order.  Used by the parity tests and smoke(): 'fixed seeds' parity is
        injected-noise parity, the torch/NumPy generators cannot be reproduced on the device."""
        self.steps += 1
        self.flush()
        buffer.flush()
        self._img_dirty = True
        self._inject = dict(idx=list(idx), eps=list(eps))
        try:
            self._body(buffer, batch_size, False)
        finally:
            self._inject = None
        return self.core.info()

    def _train_eager(self, buffer, B):
        self.flush()
        buffer.flush()
        self._img_dirty = True
        self._body(buffer, B, False)
        return self.core.info()

    def _train_graph_dp(self, buffer, B):
        """world_size > 1: the launches between two gradient all-reduces are captured as separate hipGraphs
        (7 for vlsac) and replayed around eager RCCL all-reduces."""
        import torch.distributed as dist
        buffer.flush()
        buffer.size_dev()
        key = self._graph_cache_key(buffer, B)
        if self._graph is None or self._graph_key != key:
            with _no_gc():          # (a cyclic-GC pass inside a capture may run destructors that touch the device: see _no_gc)
                self._sample_into(buffer, B, 'warm', 0, False)
                idx_keys, eps_specs = self._plan(B)            # allocate the pools outside any capture (no launch: the train
                self._buf('pool_idx', (len(idx_keys) * B,), torch.int32)   # prologue would count a step and draw from the generator)
                self._buf('pool_eps', (sum(int(np.prod(sh)) for _, sh in eps_specs),))
                torch.cuda.synchronize()
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                self._seg_capture_colls = self.capture_collectives and dist.is_initialized() and dist.get_backend() == 'nccl'
                self._n_captured_colls = 0
                if self._seg_capture_colls:
                    # the communicator must exist before the capture begins (its lazy creation allocates: not permitted while capturing);
                    # every rank builds its graph at the same train() call, so this is a matched collective
                    dist.all_reduce(torch.zeros(1, device=self.core.device))
                    torch.cuda.synchronize()
                self._hist = not _sw.off('info_history')      # see _train_graph
                self.core.history(self._hist)
                try:
                    with torch.cuda.stream(s):
                        first = torch.cuda.CUDAGraph()
                        # thread-local capture mode: the process group's watchdog thread may query events while we capture
                        first.capture_begin(capture_error_mode='thread_local')
                        self._seg = ([], first)
                        try:
                            self._body(buffer, B, True)
                            segs, cur = self._seg
                            self._seg = None
                            cur.capture_end()
                            segs.append(('graph', cur))
                        finally:
                            self.core.history(False)
                            self._abort_open_capture()
                    torch.cuda.current_stream().wait_stream(s)
                    torch.cuda.synchronize()
                except Exception as e:
                    self._raise_capture_hint(e)
                self._graph, self._graph_key = segs, key
                self._hist_n = self.core.history_seq() if self._hist else 0
                self._fresh_dp_graph = True
        try:
            for kind, x in self._graph:
                if kind == 'graph':
                    x.replay()
                else:
                    x()
            if self._fresh_dp_graph:                  # the first replay of a newly built graph is checked to the end (once per capture)
                self._fresh_dp_graph = False
                torch.cuda.synchronize()
        except Exception as e:
            self._raise_capture_hint(e)
        if self._hist:
            return self._history_info()
        return self.core.info()

    # ---- pipelined graph mode ---------------------------------------------------------------------------------------
    # train(t) = feature steps(t) -> critic(t) -> actor(t).  Feature steps read and write (encoder, decoder, f, f_target); critic
    # and actor read f_target / the last minibatch / their noise and write (critic, critic_target, actor, log_alpha).  With those
    # reads snapshotted (rlrep_defer_snapshot) the critic and actor steps of train(t) are independent of the feature steps of
    # train(t+1), and the steady-state graph runs them as TWO CONCURRENT BRANCHES: the dependent-launch chain per train() drops
    # from 71 launches to max(48 feature, 25 critic+actor).  Every parameter sees exactly the updates, in exactly the order, of
    # the sequential train(); `flush()` (called by everything that looks at the critic / actor: select_action, checkpoints,
    # reading a returned info dict, the eager step methods) runs the one pending critic+actor pair.
    def _feature_part(self, buffer, B, snap_set=None):
        self._pool = None
        self._next_key = {}
        self._fill_pools(buffer, B, True)           # rlrep_train_prologue: steps += 1, pools, first gather
        self._early_key = None                      # both policy forwards belong to the deferred branch
        n = self._feature_iters()
        for i in range(n):
            if snap_set is not None and i == n - 1:
                # the snapshot for the deferred critic / actor chain rides in this last step's optimizer launch (rlrep_defer_arm)
                self.core.defer_arm(self._pool['eps_crit'], self._pool['eps_act'], snap_set)
            self._feature_once(buffer, B, i, True)
        return self._pool['eps_crit'], self._pool['eps_act']

    def _stamp(self, tag):
        """RLREP_ENABLE=stamp (diagnostics, tools/exp/chain_stamps.py): a one-thread launch that dates this point of the chain on the device."""
        if not self._stamp_on:
            return
        from rlrep_amd._lib import lib as _l, check as _check
        if getattr(self, '_stamp_ring', None) is None:
            self._stamp_ring = torch.zeros(1 + 8192, dtype=torch.int64, device=self.core.device)
        _check(_l.rlrep_debug_stamp(self._stamp_ring.data_ptr(), 8192, int(tag), torch.cuda.current_stream().cuda_stream), 'debug_stamp')

    def _wait_set_free(self, P, k, s_f):
        """Snapshot set k is about to be overwritten by the feature chain of this call: the critic / actor chain that read it last
        (train t-nset) must have finished.  Waited for on the HOST: a wait packet in the feature stream costs ~12 us of its critical path
        (2 816 -> 2 917 train()/s); the host then runs at most nset calls ahead of the device.  With nset = 2 that wait ends about one
        graph-launch latency before the running feature chain does and the feature queue idles between calls; the third set
        (RLREP_ENABLE=defer_sets=N, default 3) removes it."""
        P['ev_ca'][k].synchronize()

    def _order_buffer_writes(self):
        """ReplayBuffer hook: the caller's stream is about to overwrite ring rows / the size scalar that the feature chain of the
        pipelined train() in flight may still be sampling from on its own stream -- make the caller's stream wait for that chain."""
        P = self._pipe
        if P is not None and self._pending == 2 and P.get('last_f') is not None:
            cur = _current_stream()
            if P.get('s_f') is not None and cur == P['s_f']:
                return                      # the write is issued ON the feature stream: ordered behind the chain by the stream itself
            cur.wait_event(P['last_f'])

    def _train_graph_pipelined(self, buffer, B):
        self._hook_buffer(buffer)
        P0 = self._pipe
        if P0 is not None and P0.get('mode') == 2 and self._pending == 2:
            # rows staged by add() since the last call and the size scalar are written ON THE FEATURE STREAM: behind the chain that may
            # still be sampling from the ring, in front of the one this call launches -- ordered by the stream, no cross-stream wait.  (On the
            # caller's stream the write had to wait for the chain in flight: a barrier packet parked in the caller's queue for most of every
            # period, which taxes both chains' launches -- DESIGN.md 5.4; add() + train() in a loop: 2 650 -> 3 360 train()/s, tools/exp/add_train_loop.py.)
            e0 = getattr(buffer, 'device_epoch', None)
            with _on_stream(P0['s_f']):
                buffer.flush()
                buffer.size_dev()
            cur_id = _raw_stream()
            if e0 is not None and P0.get('seen') == (e0, cur_id):
                P0['seen'] = (buffer.device_epoch, cur_id)        # our own writes need no wait for the caller's stream
        else:
            buffer.flush()
            buffer.size_dev()
        key = self._graph_cache_key(buffer, B)
        c = self.core
        if self._pipe is None or self._pipe['key'] != key:
            with _no_gc(), self._managed_images():          # (a cyclic-GC pass inside a capture may run destructors that touch the device: see _no_gc)
                self.flush()
                self._sample_into(buffer, B, 'warm', 0, False)      # sizes the library's tables for B outside any capture
                idx_keys, eps_specs = self._plan(B)
                self._buf('pool_idx', (len(idx_keys) * B,), torch.int32)
                self._buf('pool_eps', (sum(int(np.prod(sh)) for _, sh in eps_specs),))
                torch.cuda.synchronize()
                mode = int(os.environ.get('RLREP_PIPELINE', '2'))
                s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
                P = dict(key=key, mode=mode, t=0)
                if mode == 1:
                    # one graph per train(): the two branches inside it (snapshot set 0 only).  In-graph branches cost ~2.6 us per launch
                    # pair on this runtime (tools/exp/twochains.hip); kept as the single-stream form.
                    first, steady, tail = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                    with torch.cuda.graph(first, stream=s1):
                        ec, ea = self._feature_part(buffer, B)
                        c.defer_snapshot(ec, ea, 0)
                        c.end_train()
                    with torch.cuda.graph(steady, stream=s1):
                        fork = torch.cuda.Event()
                        fork.record()
                        s2.wait_event(fork)
                        c.deferred_critic_actor(0)                      # branch 1 (s1): critic + actor of the previous train()
                        with torch.cuda.stream(s2):                      # branch 2 (s2): this train()'s feature steps
                            ec, ea = self._feature_part(buffer, B)
                            join = torch.cuda.Event()
                            join.record()
                        torch.cuda.current_stream().wait_event(join)
                        c.defer_snapshot(ec, ea, 0)
                        c.end_train()
                    with torch.cuda.graph(tail, stream=s1):
                        c.deferred_critic_actor(0)
                    P.update(first=first, steady=steady, tail=tail)
                else:
                    # two streams that were TIMED to be concurrent; train(t) uses snapshot set t % nset:
                    #   stream F : [feature steps(t) + snapshot(t -> set)]            after the critic/actor pair of t-2 (same set)
                    #   stream CA: [critic + actor(t) from set]                       after snapshot(t)
                    from rlrep_amd._lib import lib as _l, front_end_counts as _fe
                    fs, ca = [], []
                    n0, f0 = _l.rlrep_launch_counter(), _fe()
                    nset = min(c.defer_supported(), max(2, int(_sw.opt('defer_sets', '3'))))
                    for k in range(nset):
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g, stream=s1):
                            self._stamp(1)
                            ec, ea = self._feature_part(buffer, B, snap_set=k)
                            c.defer_snapshot(ec, ea, k)
                            c.end_train()
                            self._stamp(2)
                        fs.append(g)
                    n1 = _l.rlrep_launch_counter()
                    for k in range(nset):
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g, stream=s1):
                            self._stamp(3)
                            c.deferred_critic_actor(k)
                            self._stamp(4)
                        ca.append(g)
                    P['launches'] = ((n1 - n0) // nset, (_l.rlrep_launch_counter() - n1) // nset)      # kernels in the feature / critic+actor graph
                    f1 = _fe()
                    P['front_ends'] = {k: (f1[k] - f0[k]) // nset for k in f1}      # 16-row tile engine launches of one train() per front end
                    s_ca, s_f = self._stream_pair()
                    P.update(fs=fs, ca=ca, s_ca=s_ca, s_f=s_f, nset=nset, ev_snap=[torch.cuda.Event() for _ in range(nset)],
                             ev_ca=[torch.cuda.Event() for _ in range(nset)], used=[False] * nset)
                self._pipe = P
        P = self._pipe
        if self._prepare_only:
            return None
        self._sync_images()
        if P['mode'] == 1:
            (P['steady'] if self._pending else P['first']).replay()
            self._pending = True
            return self.core.info(lazy_source=self._flushed_metrics)
        k = P['t'] % P['nset']
        P['t'] += 1
        s_ca, s_f = P['s_ca'], P['s_f']
        cur = _current_stream()
        # replay rows staged by ReplayBuffer.add() land on the caller's stream: order the feature chain after them -- but only when the
        # buffer has enqueued something since the last call (or the chains were idle): the event record + wait pair is ~5 us of the
        # feature chain's critical path, and a back-to-back train() loop has nothing to wait for
        seen = (getattr(buffer, 'device_epoch', None), cur.cuda_stream)
        if not self._pending or seen[0] is None or P.get('seen') != seen:
            s_f.wait_stream(cur)
            P['seen'] = seen
        # (the critic / actor stream needs no wait of its own for the caller's stream: its chain starts behind the snapshot event, which the
        #  feature stream records after a chain that has just been ordered behind the caller's stream)
        with _on_stream(s_f):
            if P['used'][k]:
                self._wait_set_free(P, k, s_f)
            P['fs'][k].replay()
            P['ev_snap'][k].record(s_f)
            P['last_f'] = P['ev_snap'][k]
        with _on_stream(s_ca):
            s_ca.wait_event(P['ev_snap'][k])
            P['ca'][k].replay()
            P['ev_ca'][k].record(s_ca)
        P['used'][k] = True
        self._pending = 2
        return self.core.info(lazy_source=self._flushed_metrics, early=(self.FEATURE_KEYS, self._feature_metrics_of(P['ev_snap'][k])))

    # ---- data parallel + deferred critic / actor chain ---------------------------------------------------------------------------
    # The two launch chains of the pipelined mode, each cut into hipGraph segments at its gradient all-reduces (feature chain: 4,
    # critic/actor chain: 2).  Each chain has its OWN process group (communicator + RCCL stream): within a communicator the issue
    # order is the program order of one chain, identical on every rank; the two communicators never wait for each other, so the
    # feature chain's all-reduces of train(t+1) are not queued behind the critic / actor ones of train(t); and BOTH chains of a call are
    # issued by that call -- flush() (reading a returned info dict, select_action, ...) only waits, it never issues a collective, so a
    # rank that reads its metrics while the others do not cannot desynchronise the ranks.

    def _capture_segments(self, fn):
        first = torch.cuda.CUDAGraph()
        first.capture_begin(capture_error_mode='thread_local')
        self._seg = ([], first)
        try:
            fn()
            segs, cur = self._seg
            self._seg = None
            cur.capture_end()
            segs.append(('graph', cur))
        finally:
            self._abort_open_capture()
        return segs

    def _abort_open_capture(self):
        """An exception inside a segmented capture: end the open segment so that the stream leaves capture mode (the caller's
        synchronize() would otherwise raise on a still-capturing stream), then forget the segments."""
        if self._seg is not None:
            _, cur = self._seg
            self._seg = None
            try:
                cur.capture_end()
            except Exception:
                pass

    def _train_graph_dp_pipelined(self, buffer, B):
        if self._pg_ca is None:
            import torch.distributed as dist
            self._pg_ca = dist.new_group()             # collective: every rank reaches its first pipelined train() (same program)
        self._hook_buffer(buffer)
        buffer.flush()
        buffer.size_dev()
        key = self._graph_cache_key(buffer, B)
        c = self.core
        if self._pipe is None or self._pipe['key'] != key:
            with _no_gc():          # (a cyclic-GC pass inside a capture may run destructors that touch the device: see _no_gc)
                self.flush()
                self._sample_into(buffer, B, 'warm', 0, False)
                idx_keys, eps_specs = self._plan(B)
                self._buf('pool_idx', (len(idx_keys) * B,), torch.int32)
                self._buf('pool_eps', (sum(int(np.prod(sh)) for _, sh in eps_specs),))
                import torch.distributed as dist
                self._seg_capture_colls = self.capture_collectives and dist.is_initialized() and dist.get_backend() == 'nccl'
                if self._seg_capture_colls:          # both communicators exist before the captures begin (see _train_graph_dp)
                    dist.all_reduce(torch.zeros(1, device=c.device))
                    dist.all_reduce(torch.zeros(1, device=c.device), group=self._pg_ca)
                torch.cuda.synchronize()
                cap = torch.cuda.Stream()
                cap.wait_stream(torch.cuda.current_stream())
                fs, cs = [], []
                # captured collectives: a call is 2 + 2 graph launches, the host has time to wait for the set's last reader itself (no wait parked
                # in the feature queue) and three snapshot sets keep it a call ahead; segments around eager collectives: 14 items
                # per call, two sets and a stream wait as before
                nset = 3 if (self._seg_capture_colls and c.defer_supported() >= 3) else 2
                with torch.cuda.stream(cap):
                    for k in range(nset):
                        def feature_chain(k=k):
                            ec, ea = self._feature_part(buffer, B)
                            c.defer_snapshot(ec, ea, k)
                            c.end_train()
                        fs.append(self._capture_segments(feature_chain))

                        def ca_chain(k=k):
                            c.deferred_part(k, 0); self._allreduce(1, pg=self._pg_ca); c.deferred_part(k, 1)
                            c.deferred_part(k, 2); self._allreduce(2, True, pg=self._pg_ca); c.deferred_part(k, 3)
                        cs.append(self._capture_segments(ca_chain))
                torch.cuda.current_stream().wait_stream(cap)
                torch.cuda.synchronize()
                s_ca, s_f = self._stream_pair()
                self._pipe = dict(key=key, mode=3, t=0, nset=nset, host_wait=nset == 3, fs=fs, cs=cs, s_ca=s_ca, s_f=s_f,
                                  ev_snap=[torch.cuda.Event() for _ in range(nset)], ev_ca=[torch.cuda.Event() for _ in range(nset)], used=[False] * nset)
        P = self._pipe
        k = P['t'] % P['nset']
        P['t'] += 1
        s_ca, s_f = P['s_ca'], P['s_f']
        cur = _current_stream()
        s_f.wait_stream(cur)
        if not self._pending:
            s_ca.wait_stream(cur)
        if P['used'][k]:
            if P.get('host_wait'):
                P['ev_ca'][k].synchronize()                # (train t-3: long done)
            else:
                s_f.wait_event(P['ev_ca'][k])              # the pair that read this snapshot set last (train t-2); a stream wait here: the
                                                           # host, which issues 14 items per call in this form, must keep its run-ahead
        with _on_stream(s_f):
            for kind, x in P['fs'][k]:
                x.replay() if kind == 'graph' else x()
            P['ev_snap'][k].record(s_f)
        P['last_f'] = P['ev_snap'][k]
        with _on_stream(s_ca):
            s_ca.wait_event(P['ev_snap'][k])
            for kind, x in P['cs'][k]:
                x.replay() if kind == 'graph' else x()
            P['ev_ca'][k].record(s_ca)
        P['used'][k] = True
        self._pending = 2
        return self.core.info(lazy_source=self._flushed_metrics, early=(self.FEATURE_KEYS, self._feature_metrics_of(P['ev_snap'][k])))

    @staticmethod
    def _graph_cache_key(buffer, B):
        """What a captured train() bakes in: the ring and size-scalar device addresses and the batch size (id(buffer) can be reused by
        a new ReplayBuffer after the old one is collected)."""
        return (buffer.ring.data_ptr(), buffer.size_dev().data_ptr(), int(buffer.ring.shape[0]), B)

    def _hook_buffer(self, buffer):
        hooks = getattr(buffer, 'before_device_write_hooks', None)
        if hooks is not None and self._order_buffer_writes not in hooks:
            hooks.append(self._order_buffer_writes)
        self._held_buffer = buffer          # the cached graphs read its ring: keep it alive as long as they are

    def _flushed_metrics(self):
        self.flush()
        return self.core.metrics_tensor().clone()

    def _feature_metrics_of(self, ev_feature_done):
        """Source of the feature-step losses of a pipelined train(): they are final once its feature chain has ended (the event), while its
        critic / actor chain may still be running -- reading only those keys does not end the overlap with the next call."""
        def fetch():
            ev_feature_done.synchronize()
            return self.core.metrics_tensor().clone()
        return fetch

    def _prefer_sequential(self):
        """Per-call choice between the two-chain and the one-graph form of train() (see __init__): has the caller been waiting for the critic /
        actor pair between the calls?"""
        if not self._adaptive or self._pipeline_mode == 1:
            return False
        if self._looked:
            self._n_looked, self._n_b2b = min(self._n_looked + 1, 8), 0
        else:
            self._n_b2b += 1
            if self._n_b2b >= 2:
                self._n_looked = 0
        return self._n_looked >= 3

    def flush(self):
        """Finish the critic + actor steps of the last pipelined train() (no-op otherwise)."""
        self._looked = True
        self.core.exchange_check()               # a peer that never arrived (bounded waits of the in-launch gradient exchange): raise, do not train on
        if self._pending == 2:                         # two-stream forms: the pair is already in flight on its own streams
            self._pending = False
            P = self._pipe
            # Waited for on the HOST (every caller of flush() is about to read results anyway).  A stream-level wait here -- a barrier
            # packet parked in the caller's hardware queue until both chains are done -- slows the launches still queued on the two
            # chains' queues by ~12 % for as long as it is parked (tools/exp/window_stamps.py: the last two feature chains of a window
            # 283 -> 315-335 us, the last critic / actor chain 193 -> 227 us; the same effect that made the set-reuse wait a host wait).
            # The critic / actor chain of the last call is the last thing in flight: its feature chain ended before it started.
            P['ev_ca'][(P['t'] - 1) % P['nset']].synchronize()
            self.core.exchange_check()           # ... and again behind the wait: a timeout raised by the launches that were still running (advisor r05)
        elif self._pending:
            self._pending = False
            self._pipe['tail'].replay()

    def _train_graph(self, buffer, B):
        buffer.flush()
        buffer.size_dev()
        key = self._graph_cache_key(buffer, B)
        if self._graph is None or self._graph_key != key:
            with _no_gc():          # (a cyclic-GC pass inside a capture may run destructors that touch the device: see _no_gc)
                # size the library's tables for B outside the capture (it re-uploads them with blocking copies
                # when the batch size changes), then capture the whole train() into one hipGraph
                self._sample_into(buffer, B, 'warm', 0, False)
                torch.cuda.synchronize()
                from rlrep_amd._lib import lib as _l, front_end_counts as _fe
                s = torch.cuda.Stream()
                g = torch.cuda.CUDAGraph()
                n0, f0 = _l.rlrep_launch_counter(), _fe()
                # the call's metrics are filed in the library's history ring by the last launch of the graph (rlrep_history) and fetched when the
                # returned dict is read: no snapshot launch per call (sac: 10 120 -> 10 600 train()/s).  RLREP_DISABLE=info_history: a clone per call.
                self._hist = not _sw.off('info_history')
                self.core.history(self._hist)
                try:
                    with self._managed_images(), torch.cuda.graph(g, stream=s):
                        self._body(buffer, B, True)
                finally:
                    self.core.history(False)
                self._graph, self._graph_key = g, key
                self._graph_launches = _l.rlrep_launch_counter() - n0
                f1 = _fe()
                self._graph_front_ends = {k: f1[k] - f0[k] for k in f1}
                self._hist_n = self.core.history_seq() if self._hist else 0           # (synchronises; once per capture)
        if self._prepare_only:
            return None
        self._sync_images()
        self._graph.replay()
        if self._hist:
            return self._history_info()
        return self.core.info()
