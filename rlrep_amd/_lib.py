"""ctypes binding of librlrep_hip.so (the C ABI declared in include/rlrep.h).

There is NO fallback: if the HIP library is missing or fails to load, importing this module raises.
`import torch` must happen first so that the library resolves libamdhip64.so.7 to the HIP runtime
torch already loaded (device pointers and streams are shared with torch).
"""
import ctypes as C
import os
import re

import torch  # noqa: F401  (must precede the CDLL load, see docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('RLREP_LIB') or os.path.join(_HERE, 'lib', 'librlrep_hip.so')     # RLREP_LIB: A/B builds of the same ABI
HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'rlrep.h')

ALG = {'sac': 0, 'vlsac': 1, 'ctrlsac': 2, 'spedersac': 3, 'diffsrsac': 4}
ARENA_PARAM, ARENA_TARGET = 0, 1
GRAD_TAIL = 256
FLAG_NO_FEATURE_TARGET = 1


class Dims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        'alg', 'state_dim', 'action_dim', 'hidden_dim', 'actor_hidden_dim', 'feature_dim', 'vae_hidden_dim',
        'phi_hidden_dim', 'phi_hidden_depth', 'mu_hidden_dim', 'mu_hidden_depth', 'num_noise', 'max_batch', 'rank', 'flags', 'world_size')]


class Hyper(C.Structure):
    _fields_ = [('lr_feature', C.c_float), ('lr_critic', C.c_float), ('lr_actor', C.c_float),
                ('discount', C.c_float), ('tau', C.c_float), ('feature_tau', C.c_float),
                ('target_entropy', C.c_float), ('sigma_scale', C.c_float),
                ('target_update_period', C.c_int32), ('extra_feature_steps', C.c_int32),
                ('learn_alpha', C.c_int32), ('world_size', C.c_int32),
                ('beta1', C.c_float), ('beta2', C.c_float), ('adam_eps', C.c_float), ('critic_reg_lambda', C.c_float)]


class TensorDesc(C.Structure):
    _fields_ = [('name', C.c_char * 72), ('arena', C.c_int32), ('group', C.c_int32), ('offset', C.c_int64),
                ('rows', C.c_int32), ('cols', C.c_int32)]


class LayoutInfo(C.Structure):
    _fields_ = [('param_floats', C.c_int64), ('target_floats', C.c_int64), ('grad_floats', C.c_int64),
                ('workspace_bytes', C.c_int64), ('group_offset', C.c_int64 * 4), ('group_floats', C.c_int64 * 4),
                ('n_tensors', C.c_int32), ('n_metrics', C.c_int32), ('exchange_floats', C.c_int64)]


class Arenas(C.Structure):
    _fields_ = [('param_dev', C.c_void_p), ('target_dev', C.c_void_p), ('grad_dev', C.c_void_p),
                ('exp_avg_dev', C.c_void_p), ('exp_avg_sq_dev', C.c_void_p), ('workspace_dev', C.c_void_p),
                ('alpha_state_dev', C.c_void_p)]


class Batch(C.Structure):
    _fields_ = [('state_dev', C.c_void_p), ('action_dev', C.c_void_p), ('reward_dev', C.c_void_p),
                ('next_state_dev', C.c_void_p), ('done_dev', C.c_void_p), ('batch', C.c_int32)]


def declared_symbols():
    """Every function name include/rlrep.h declares (used by the CPU test that checks the exports)."""
    src = open(HEADER_PATH).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(rlrep_[a-z_0-9]+)\s*\(', src)))


def _load():
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            f'(hipcc --offload-arch=gfx950).  rlrep_amd has no CPU fallback.')
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, u64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float
    P = C.POINTER
    sig = {
        'rlrep_abi_version': (i32, []),
        'rlrep_last_error': (C.c_char_p, []),
        'rlrep_layout': (i32, [P(Dims), P(LayoutInfo), P(TensorDesc), i32]),
        'rlrep_metric_names': (i32, [i32, vp, i32]),
        'rlrep_agent_create': (i32, [P(Dims), P(Hyper), P(Arenas), vp, P(vp)]),
        'rlrep_agent_destroy': (None, [vp]),
        'rlrep_set_batch': (i32, [vp, i32, P(Batch), vp]),
        'rlrep_replay_row_floats': (i32, [P(Dims)]),
        'rlrep_replay_add': (i32, [vp, i64, i32, i64, vp, i64, vp]),
        'rlrep_replay_add_sized': (i32, [vp, i64, i32, i64, vp, i64, vp, i32, vp]),
        'rlrep_replay_sample': (i32, [vp, i32, vp, vp, i32, vp]),
        'rlrep_fill_indices': (i32, [vp, i64, i32, u64, u64, vp]),
        'rlrep_fill_normal': (i32, [vp, i64, f32, u64, u64, vp]),
        'rlrep_philox_raw': (i32, [vp, vp, i64, vp]),
        'rlrep_fill_indices_dev': (i32, [vp, i64, vp, u64, u64, vp, vp]),
        'rlrep_fill_normal_dev': (i32, [vp, i64, f32, u64, u64, vp, vp]),
        'rlrep_steps_dev': (vp, [vp]),
        'rlrep_group_cfg_dev': (vp, [vp]),
        'rlrep_feature_step': (i32, [vp, vp, vp, vp]),
        'rlrep_prefetch_policy': (i32, [vp, vp]),
        'rlrep_prefetch_policy_early': (i32, [vp, vp, vp]),
        'rlrep_prefetch_batch': (i32, [vp, vp, vp, i32]),
        'rlrep_prefetch_batch_slot': (i32, [vp, i32, vp, vp, i32]),
        'rlrep_train_prologue': (i32, [vp, vp, vp, vp, i64, vp, i64, u64, u64, u64, i32, vp]),
        'rlrep_critic_step': (i32, [vp, vp, vp]),
        'rlrep_actor_alpha_step': (i32, [vp, vp, vp]),
        'rlrep_update_target': (i32, [vp, vp]),
        'rlrep_begin_train': (i32, [vp, vp]),
        'rlrep_feature_backward': (i32, [vp, vp, vp, vp]),
        'rlrep_feature_apply': (i32, [vp, vp]),
        'rlrep_critic_backward': (i32, [vp, vp, vp]),
        'rlrep_critic_apply': (i32, [vp, vp]),
        'rlrep_actor_backward': (i32, [vp, vp, vp]),
        'rlrep_actor_apply': (i32, [vp, vp]),
        'rlrep_feature_exchange_count': (i32, [vp]),
        'rlrep_feature_exchange': (i32, [vp, i32, P(i32), P(vp), P(i64), P(i64)]),
        'rlrep_feature_backward_part': (i32, [vp, i32, vp, vp, vp]),
        'rlrep_defer_supported': (i32, [vp]),
        'rlrep_defer_snapshot': (i32, [vp, i32, vp, vp, vp]),
        'rlrep_defer_arm': (i32, [vp, i32, vp, vp]),
        'rlrep_deferred_critic_actor': (i32, [vp, i32, vp]),
        'rlrep_deferred_part': (i32, [vp, i32, i32, vp]),
        'rlrep_end_train': (i32, [vp]),
        'rlrep_sync_frozen': (i32, [vp, vp]),
        'rlrep_actor_forward': (i32, [vp, vp, i32, vp, f32, f32, vp, vp]),
        'rlrep_select_action': (i32, [vp, vp, i32, i32, u64, u64, f32, f32, vp, i32, vp]),
        'rlrep_images_managed': (i32, [vp, i32]),
        'rlrep_refresh_images': (i32, [vp, vp]),
        'rlrep_feature_chain_next': (i32, [vp]),
        'rlrep_comm_create': (i32, [i32, i32, i64, i64, P(vp)]),
        'rlrep_comm_handle_bytes': (i32, []),
        'rlrep_comm_handle': (i32, [vp, vp, i32]),
        'rlrep_comm_connect': (i32, [vp, vp]),
        'rlrep_comm_connect_local': (i32, [vp, P(vp)]),
        'rlrep_comm_set_timeout': (i32, [vp, i64]),
        'rlrep_comm_arena': (vp, [vp]),
        'rlrep_comm_scratch': (vp, [vp]),
        'rlrep_comm_attach': (i32, [vp, vp, i64, i64, P(i32)]),
        'rlrep_comm_allreduce': (i32, [vp, i64, i64, vp, i32, i64, vp]),
        'rlrep_comm_allgather': (i32, [vp, i64, i64, vp]),
        'rlrep_comm_probe_fill': (i32, [vp, i64, i64, i32, vp]),
        'rlrep_comm_probe_value': (f32, [i32, i32, i64]),
        'rlrep_comm_probe_slots': (i32, [vp, i64, i32, vp, i64, vp]),
        'rlrep_comm_status': (i32, [vp, P(C.c_uint32), i32]),
        'rlrep_comm_fine_grained': (i32, [vp]),
        'rlrep_comm_debug_preset': (i32, [vp, i32, i32]),
        'rlrep_comm_destroy': (None, [vp]),
        'rlrep_stage_count': (i32, [vp, i32]),
        'rlrep_stage_name': (C.c_char_p, [vp, i32, i32]),
        'rlrep_run_stage': (i32, [vp, i32, i32, vp]),
        'rlrep_stage_info': (i32, [vp, i32, i32, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        'rlrep_gemm': (i32, [i32, i32, i32, vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, i32, vp, i32, i32, vp, i64, vp]),
        'rlrep_gemm_plan': (i32, [i32] * 8 + [P(i32)] * 5),
        'rlrep_nc_fwd_plan': (i32, [i32] * 4 + [P(i32)] * 3),
        'rlrep_chain_status': (i32, [vp, P(C.c_uint32), vp]),
        'rlrep_build_flags': (i32, []),
        'rlrep_debug_stamp': (i32, [vp, i32, i32, vp]),
        'rlrep_history': (i32, [vp, i32]),
        'rlrep_history_dev': (i32, [vp, P(vp), P(vp), P(i32), P(i32), P(i32)]),
        'rlrep_metrics_dev': (vp, [vp]),
        'rlrep_last_launch_count': (i32, [vp]),
        'rlrep_launch_counter': (i64, []),
        'rlrep_front_end_counts': (i32, [P(i64)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)       # AttributeError if the library does not export it
        fn.restype, fn.argtypes = res, args
    if lib.rlrep_abi_version() != 4:
        raise RuntimeError('librlrep_hip.so ABI version mismatch')
    return lib, sig


lib, SIGNATURES = _load()


def has_experiments():
    """True if the loaded library carries the opt-in engines (built with RLREP_BUILD_EXPERIMENTS=1)."""
    return bool(lib.rlrep_build_flags() & 1)


def check(rc, what=''):
    if rc != 0:
        msg = lib.rlrep_last_error()
        raise RuntimeError(f'rlrep {what} failed ({rc}): {msg.decode() if msg else ""}')


def front_end_counts():
    """Launches of the 16-row tile engine per front end since the library was loaded: dict(fast, fast4, fastpre, record)."""
    out = (C.c_int64 * 4)()
    check(lib.rlrep_front_end_counts(out), 'front_end_counts')
    return dict(zip(('fast', 'fast4', 'fastpre', 'record'), (int(v) for v in out)))
