"""ctypes binding of librainbow_hip.so (C ABI declared in include/rainbow_hip.h).

The library is hand-written HIP for gfx950; there is NO fallback: if the shared object is
missing or does not load, importing the hot path raises.  `declare()` only attaches
argtypes/restypes and is reused by tests/hipemu (the host-interpreted test build of the
same sources) so both builds are checked against one signature table.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RAINBOW_AMD_LIB: another build of the SAME sources (e.g. with an experiment's -D switch) for same-box A/B runs
LIB_PATH = os.environ.get("RAINBOW_AMD_LIB") or os.path.join(_HERE, "librainbow_hip.so")

c_void_p, c_int, c_int32, c_int64, c_uint64 = C.c_void_p, C.c_int, C.c_int32, C.c_int64, C.c_uint64
c_float, c_double, c_char_p = C.c_float, C.c_double, C.c_char_p


class ReplayHeader(C.Structure):
    _fields_ = [("index", c_int64), ("full", c_int32), ("max", c_float), ("total", c_float),
                ("last_attempts", c_int32), ("last_status", c_int32), ("rng_counter", c_uint64)]


class ReplayBuffers(C.Structure):
    _fields_ = [("sum_tree_dev", c_void_p), ("tree_len", c_int64), ("tree_start", c_int64),
                ("frames_dev", c_void_p), ("timestep_dev", c_void_p), ("action_dev", c_void_p),
                ("reward_dev", c_void_p), ("nonterminal_dev", c_void_p), ("header_dev", c_void_p),
                ("window_dev", c_void_p), ("window_len", c_int32)]


class LearnerConfig(C.Structure):
    _fields_ = [("batch", c_int32), ("atoms", c_int32), ("actions", c_int32), ("history", c_int32),
                ("hidden", c_int32), ("architecture", c_int32), ("multi_step", c_int32),
                ("v_min", c_float), ("v_max", c_float), ("discount", c_double)]


class NoiseJob(C.Structure):
    _fields_ = [("opaque", c_uint64 * 32)]


class TrainStep(C.Structure):
    """rb_train_step_t (include/rainbow_hip.h)."""
    _fields_ = [("replay", c_void_p), ("batch", c_int32), ("max_attempts", c_int32), ("window_len", c_int32), ("reserved", c_int32),
                ("priority_weight", c_double), ("tree_idx_dev", c_void_p), ("actions_dev", c_void_p), ("returns_dev", c_void_p),
                ("nonterminals_dev", c_void_p), ("weights_dev", c_void_p), ("noise_job", c_void_p), ("frames_dev", c_void_p),
                ("windows_dev", c_void_p), ("loss_dev", c_void_p), ("exp_avg_dev", c_void_p), ("exp_avg_sq_dev", c_void_p),
                ("norm_dev", c_void_p), ("lr", c_double), ("beta1", c_double), ("beta2", c_double), ("eps", c_double),
                ("step", c_int64), ("max_norm", c_float), ("reserved2", c_float)]


class TensorDesc(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("offset", c_int64), ("ndim", c_int32), ("shape", c_int32 * 4)]


# name -> (restype, argtypes); every symbol include/rainbow_hip.h declares
SIGNATURES = {
    "rb_last_error": (c_char_p, []),
    "rb_abi_version": (c_int, []),
    "rb_source_hash": (c_char_p, []),
    "rb_copy_to_host": (c_int, [c_void_p, c_void_p, C.c_size_t, c_void_p]),
    "rb_copy_to_device": (c_int, [c_void_p, c_void_p, C.c_size_t, c_void_p]),
    "rb_profile_select": (c_int, [c_char_p]),
    "rb_profile_stride": (c_int, [c_int32]),
    "rb_profile_read": (c_int, [C.POINTER(c_double), C.POINTER(c_int64)]),
    "rb_profile_overhead": (c_int, [c_void_p, c_int32, C.POINTER(c_double)]),
    "rb_debug_check_guards": (c_int, [C.POINTER(c_int64), C.POINTER(c_int64)]),
    "rb_replay_create": (c_int, [C.POINTER(c_void_p), c_int64, c_int32, c_int32, c_double, c_double, c_uint64]),
    "rb_replay_destroy": (c_int, [c_void_p]),
    "rb_replay_buffers": (c_int, [c_void_p, C.POINTER(ReplayBuffers)]),
    "rb_replay_header": (c_int, [c_void_p, C.POINTER(ReplayHeader), c_void_p]),
    "rb_frame_preprocess": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "rb_replay_append": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_float, c_int32, c_void_p]),
    "rb_replay_append_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "rb_replay_find": (c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rb_replay_sample": (c_int, [c_void_p, c_int32, c_double, c_void_p, c_int32, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rb_replay_set_beta_source": (c_int, [c_void_p, c_void_p]),
    "rb_replay_failed_samples": (c_int, [c_void_p, C.POINTER(c_int64)]),
    "rb_replay_reset_failed_samples": (c_int, [c_void_p]),
    "rb_replay_dropped_updates": (c_int, [c_void_p, C.POINTER(c_int64)]),
    "rb_replay_expired_waits": (c_int, [c_void_p, C.POINTER(c_int64)]),
    "rb_replay_position": (c_int, [c_void_p, C.POINTER(c_int64), C.POINTER(c_int32)]),
    "rb_replay_sample_fused_noise": (c_int, [c_void_p, c_int32, c_double, c_void_p, c_int32, c_void_p, c_void_p, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_void_p, C.POINTER(NoiseJob), c_void_p]),
    "rb_replay_update_leaves": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    "rb_replay_update_priorities": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    "rb_replay_update_sample": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_double, c_void_p, c_int32, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rb_replay_state_at": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "rb_replay_states_at": (c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
    "rb_u8_to_unit_f32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "rb_learner_sizes": (c_int, [C.POINTER(LearnerConfig), C.POINTER(c_int64), C.POINTER(c_int64)]),
    "rb_learner_param_layout": (c_int, [C.POINTER(LearnerConfig), C.POINTER(TensorDesc), C.POINTER(c_int32)]),
    "rb_learner_noise_layout": (c_int, [C.POINTER(LearnerConfig), C.POINTER(TensorDesc), C.POINTER(c_int32)]),
    "rb_learner_create": (c_int, [C.POINTER(c_void_p), C.POINTER(LearnerConfig), c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_uint64]),
    "rb_learner_destroy": (c_int, [c_void_p]),
    "rb_learner_reset_noise": (c_int, [c_void_p, c_int32, c_void_p, c_void_p]),
    "rb_learner_noise_job": (c_int, [c_void_p, c_int32, C.POINTER(NoiseJob)]),
    "rb_learner_noise_draws": (c_int64, [C.POINTER(LearnerConfig)]),
    "rb_learner_act": (c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "rb_learner_act_wait": (c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, C.POINTER(c_int32), C.POINTER(c_float), c_void_p]),
    "rb_learner_act_batch": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "rb_learner_learn": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p]),
    "rb_learner_learn_windows": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p]),
    "rb_learner_zero_copy_ok": (c_int, [c_void_p]),
    "rb_learner_clip_grad": (c_int, [c_void_p, c_float, c_void_p, c_void_p]),
    "rb_learner_clip_adam": (c_int, [c_void_p, c_float, c_void_p, c_void_p, c_double, c_double, c_double, c_double,
                                     c_int64, c_void_p, c_void_p]),
    "rb_learner_clip_adam_deferred": (c_int, [c_void_p, c_float, c_void_p, c_void_p, c_double, c_double, c_double, c_double,
                                              c_int64, c_void_p, c_void_p]),
    "rb_learner_attach_pending": (c_int, [c_void_p, c_void_p, c_int32, c_void_p]),
    "rb_learner_pending_launched": (c_int, [c_void_p]),
    "rb_learner_train_step": (c_int, [c_void_p, C.POINTER(TrainStep), c_void_p]),
    "rb_learner_set_flags": (c_int, [c_void_p, c_int32]),
    "rb_learner_flush": (c_int, [c_void_p, c_void_p]),
    "rb_learner_set_step_counter": (c_int, [c_void_p, c_void_p]),
    "rb_learner_set_priority_sink": (c_int, [c_void_p, c_void_p, c_void_p]),
    "rb_learner_priority_written": (c_int, [c_void_p]),
    "rb_learner_grads_modified": (c_int, [c_void_p]),
    "rb_learner_exchange_layout": (c_int, [c_void_p, C.POINTER(c_int64), C.POINTER(c_int64), C.POINTER(c_int64)]),
    "rb_learner_set_exchange": (c_int, [c_void_p, c_int32, c_void_p, c_void_p]),
    "rb_learner_wait_factors": (c_int, [c_void_p, c_void_p]),
    "rb_learner_finish_grads": (c_int, [c_void_p, c_void_p]),
    "rb_comm_available": (c_int, []),
    "rb_comm_unique_id": (c_int, [c_void_p]),
    "rb_comm_create": (c_int, [C.POINTER(c_void_p), c_void_p, c_int32, c_int32]),
    "rb_comm_destroy": (c_int, [c_void_p]),
    "rb_learner_exchange_rccl": (c_int, [c_void_p, c_void_p, c_void_p]),
    "rb_learner_train_step_dist": (c_int, [c_void_p, C.POINTER(TrainStep), c_void_p, c_void_p]),
    "rb_learner_sync_target": (c_int, [c_void_p, c_void_p]),
    "rb_learner_get_rng": (c_int, [c_void_p, C.POINTER(c_uint64), C.POINTER(c_uint64), c_void_p]),
    "rb_learner_set_rng": (c_int, [c_void_p, c_uint64, c_uint64, c_void_p]),
    "rb_learner_debug_read": (c_int, [c_void_p, c_int32, c_void_p, c_void_p]),
}


def declare(lib, strict=True):
    """Attach restype/argtypes for every declared symbol; raises if one is missing."""
    missing = []
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    if missing and strict:
        raise ImportError("librainbow: missing C-ABI symbols: %s" % ", ".join(missing))
    return lib


LEARNER_FUSE_FC_H_DW, LEARNER_WRITE_FUSED_GRADS, LEARNER_DEFER_UPDATE, LEARNER_IMPLICIT_SIGMA = 1, 2, 4, 8


class RainbowError(RuntimeError):
    pass


def check(lib, rc):
    if rc != 0:
        msg = lib.rb_last_error()
        raise RainbowError("librainbow_hip error %d: %s" % (rc, msg.decode() if msg else "?"))


_lib = None


def load():
    """Loads librainbow_hip.so (torch first, so HIP symbols bind to the runtime PyTorch-ROCm
    already initialised — both use SONAME libamdhip64.so.7).  Fails loudly; no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "rainbow_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the hot path." % LIB_PATH)
    import torch  # noqa: F401  (loads libamdhip64 from torch/lib)
    _lib = declare(C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL))
    return _lib


def source_hash(lib):
    """Hash of the sources the loaded library was built from (see __graft_entry__.source_hash)."""
    return lib.rb_source_hash().decode()
