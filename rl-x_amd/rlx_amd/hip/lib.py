"""ctypes binding of librlxhip.so -- one Python function per entry point of
include/rlx_hip.h.  PyTorch tensors are used only as device-memory handles
(`data_ptr()`); every computation happens inside the HIP library.  There is NO
CPU fallback: if the library is missing, loading raises.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_uint32, c_void_p

import numpy as np

ACT_TANH, ACT_ELU, ACT_RELU = 0, 1, 2
THREEFRY_LEGACY, THREEFRY_PARTITIONABLE = 0, 1

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # .../rl-x_amd


def library_path():
    return os.environ.get("RLX_HIP_LIBRARY", os.path.join(_PKG_ROOT, "lib", "librlxhip.so"))


class RlxError(RuntimeError):
    pass


class MlpDesc(Structure):
    _fields_ = [("in_dim", c_int32), ("n_hidden", c_int32), ("hidden", c_int32 * 4), ("out_dim", c_int32),
                ("act", c_int32), ("ln_first", c_int32), ("has_logstd", c_int32)]


class PpoHparams(Structure):
    _fields_ = [("clip_range", c_float), ("entropy_coef", c_float), ("critic_coef", c_float),
                ("max_grad_norm", c_float), ("adam_b1", c_float), ("adam_b2", c_float), ("adam_eps", c_float),
                ("discrete_actions", c_int32), ("critic_states", c_void_p)]


class SacHparams(Structure):
    _fields_ = [(n, c_float) for n in ("gamma", "tau", "target_entropy", "log_std_min", "log_std_max", "lr_policy",
                                       "lr_critic", "lr_alpha", "adam_b1", "adam_b2", "adam_eps")] + [
        ("key_schedule", c_int32), ("critic_states", c_void_p), ("critic_next_states", c_void_p),
        ("batch_global", c_int64), ("batch_row_offset", c_int64),      # data parallel: this rank's rows of a global batch
        ("ring_states", c_void_p), ("ring_next_states", c_void_p), ("ring_actions", c_void_p), ("ring_rewards", c_void_p),
        ("ring_terminations", c_void_p), ("ring_idx1", c_void_p), ("ring_idx2", c_void_p), ("ring_nr_envs", c_int32),
        ("keep_images", c_int32)]     # the caller's per-call statement that nobody else wrote the parameter vectors (include/rlx_hip.h)


class LnMlpDesc(Structure):
    """rlx_lnmlp_desc: Dense + LayerNorm(1e-5) + SiLU per hidden layer, Dense head (FastSAC's networks)."""
    _fields_ = [("in_dim", c_int32), ("n_hidden", c_int32), ("hidden", c_int32 * 4), ("out_dim", c_int32)]


def lnmlp_desc(in_dim, hidden, out_dim):
    d = LnMlpDesc()
    d.in_dim, d.n_hidden, d.out_dim = int(in_dim), len(hidden), int(out_dim)
    for i in range(4):
        d.hidden[i] = int(hidden[i]) if i < len(hidden) else 0
    return d


class FastSacHparams(Structure):
    _fields_ = [(n, c_float) for n in ("gamma", "tau", "v_min", "v_max", "log_std_min", "log_std_max", "target_entropy", "lr_policy",
                                       "lr_critic", "lr_alpha", "weight_decay", "adam_b1", "adam_b2", "adam_eps", "max_grad_norm")] + [
        ("nr_atoms", c_int32), ("clipped_double_q", c_int32)]


class LstmPolicyDesc(Structure):
    _fields_ = [("obs_dim", c_int32), ("act_dim", c_int32), ("enc_dim", c_int32), ("lstm_hidden", c_int32),
                ("torso", c_int32 * 3), ("share_encoder", c_int32), ("cell", c_int32), ("combine", c_int32)]


CELL_LSTM, CELL_GRU = 0, 1
COMBINE_CONCAT, COMBINE_FILM = 0, 1


def lstm_policy_desc(obs_dim, act_dim, enc_dim=128, lstm_hidden=64, torso=(512, 256, 128), share_encoder=False, cell=CELL_LSTM,
                     combine=COMBINE_CONCAT):
    d = LstmPolicyDesc()
    d.obs_dim, d.act_dim, d.enc_dim, d.lstm_hidden = int(obs_dim), int(act_dim), int(enc_dim), int(lstm_hidden)
    for i in range(3):
        d.torso[i] = int(torso[i])
    d.share_encoder = int(bool(share_encoder))
    d.cell = int(cell)
    d.combine = int(combine)
    return d


def mlp_desc(in_dim, hidden, out_dim, act, ln_first, has_logstd):
    d = MlpDesc()
    d.in_dim, d.n_hidden, d.out_dim = int(in_dim), len(hidden), int(out_dim)
    for i in range(4):
        d.hidden[i] = int(hidden[i]) if i < len(hidden) else 0
    d.act, d.ln_first, d.has_logstd = int(act), int(bool(ln_first)), int(bool(has_logstd))
    return d


class ProfRow(ctypes.Structure):
    """rlx_prof_row (include/rlx_hip.h)."""
    _fields_ = [("kernel", ctypes.c_int32), ("engine", ctypes.c_int32), ("N", ctypes.c_int32), ("K", ctypes.c_int32),
                ("M", c_int64), ("launches", c_int64), ("timed", c_int64), ("ms", ctypes.c_double),
                ("flops", ctypes.c_double), ("bytes", ctypes.c_double)]


_ALLREDUCE_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_int64, c_int, c_int)
_U32P = POINTER(c_uint32)
_I64P = POINTER(c_int64)
_DESCP = POINTER(MlpDesc)
_HPP = POINTER(PpoHparams)
_F32HP = POINTER(c_float)
_SACHPP = POINTER(SacHparams)
_LDESCP = POINTER(LstmPolicyDesc)

# name -> (restype, argtypes); mirrors include/rlx_hip.h one to one
_SIGNATURES = {
    "rlx_version": (c_int, []),
    "rlx_last_error": (c_char_p, []),
    "rlx_ctx_create": (c_int, [c_int, POINTER(c_void_p)]),
    "rlx_ctx_destroy": (c_int, [c_void_p]),
    "rlx_mlp_param_count": (c_int64, [_DESCP]),
    "rlx_dbg_gemm_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int,
                                 c_void_p]),
    "rlx_dbg_l1_f32": (c_int, [c_void_p, c_int] + [c_void_p] * 7 + [c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "rlx_dbg_set_option": (c_int, [c_void_p, c_char_p, c_int]),
    "rlx_dbg_get_counter": (c_int, [c_void_p, c_char_p, _I64P]),
    "rlx_dbg_set_sac_noise": (c_int, [c_void_p, c_void_p, c_void_p]),
    "rlx_dbg_set_stamps": (c_int, [c_void_p, c_void_p]),
    "rlx_dist_rccl_path": (c_char_p, []),
    "rlx_c51_critic_loss_f32": (c_int, [c_void_p] * 11 + [c_int64, c_int, c_float, c_float, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rlx_obs_norm_update_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rlx_obs_norm_apply_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p]),
    "rlx_prof_begin": (c_int, [c_void_p]),
    "rlx_prof_union_ms": (c_int, [c_void_p, POINTER(ctypes.c_double)]),
    "rlx_prof_kernel_count": (c_int, []),
    "rlx_prof_kernel_name": (c_char_p, [c_int]),
    "rlx_prof_end": (c_int, [c_void_p, POINTER(ctypes.c_double), POINTER(ctypes.c_double), POINTER(ctypes.c_double), _I64P]),
    "rlx_prof_rows": (c_int, [c_void_p, c_void_p, c_int, POINTER(c_int)]),
    "rlx_threefry_split_host": (c_int, [_U32P, _U32P, c_int, c_int]),
    "rlx_random_bits_u32": (c_int, [c_void_p, _U32P, c_void_p, c_int64, c_int, c_void_p]),
    "rlx_normal_f32": (c_int, [c_void_p, _U32P, c_void_p, c_int64, c_int, c_void_p]),
    "rlx_permutation_i32": (c_int, [c_void_p, _U32P, c_void_p, c_int, c_int64, c_int, c_void_p]),
    "rlx_env_reset_f32": (c_int, [c_void_p, c_uint32, c_int, c_int, c_int, c_int] + [c_void_p] * 5 + [c_void_p]),
    "rlx_env_step_f32": (c_int, [c_void_p, c_uint32, c_int, c_uint32, c_int, c_int, c_int, c_int, c_float, c_float]
                         + [c_void_p] * 11 + [c_void_p]),
    "rlx_env_step_copy_f32": (c_int, [c_void_p, c_uint32, c_int, c_uint32, c_int, c_int, c_int, c_int, c_float, c_float]
                              + [c_void_p] * 12 + [c_void_p]),
    "rlx_select_columns_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int64, c_void_p]),
    "rlx_actor_critic_fwd_sample_f32": (c_int, [c_void_p, _DESCP, c_void_p, _DESCP, c_void_p, c_void_p, c_void_p, _U32P, c_int,
                                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                                c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "rlx_actor_critic_fwd_sample_discrete_f32": (c_int, [c_void_p, _DESCP, c_void_p, _DESCP, c_void_p, c_void_p, _U32P, c_int,
                                                         c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                                         c_void_p]),
    "rlx_ppo_rollout_step_supported": (c_int, [_DESCP, _DESCP]),
    "rlx_ppo_rollout_begin": (c_int, [c_void_p, _DESCP, c_void_p, _DESCP, c_void_p, c_void_p]),
    "rlx_ppo_rollout_end": (c_int, [c_void_p]),
    "rlx_ppo_rollout_step_f32": (c_int, [c_void_p, _DESCP, c_void_p, _DESCP, c_void_p, c_void_p, c_void_p, _U32P, c_int,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                         c_int, c_int, c_int, c_uint32, c_int, c_uint32, c_int, c_float, c_float]
                                 + [c_void_p] * 8 + [c_void_p]),
    "rlx_ppo_rollout_f32": (c_int, [c_void_p, _DESCP, c_void_p, _DESCP, c_void_p, c_void_p, c_void_p, _U32P, c_int,
                                    c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int,
                                    c_uint32, c_int, c_uint32, c_int, c_float, c_float] + [c_void_p] * 8 + [c_void_p]),
    "rlx_mlp_fwd_f32": (c_int, [c_void_p, _DESCP, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "rlx_ppo_next_values_f32": (c_int, [c_void_p, _DESCP] + [c_void_p] * 5 + [c_int, c_int, c_void_p]),
    "rlx_gae_f32": (c_int, [c_void_p] * 7 + [c_int, c_int, c_float, c_float, c_void_p]),
    "rlx_ppo_reduce_metrics_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_void_p]),
    "rlx_ppo_minibatch_fwd_bwd_f32": (c_int, [c_void_p, _DESCP, c_void_p, c_void_p, _DESCP, c_void_p, c_void_p,
                                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                              c_int, c_int, c_void_p, c_int, _HPP, c_void_p]),
    "rlx_grad_global_norm_f32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "rlx_clip_adam_step_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_float,
                                       c_float, c_float, c_float, c_float, c_void_p, c_void_p]),
    "rlx_sac_replay_sample_f32": (c_int, [c_void_p] * 6 + [c_int, c_int, c_int, c_void_p, c_void_p, c_int64]
                                  + [c_void_p] * 5 + [c_void_p]),
    "rlx_sac_invalidate_images": (c_int, [c_void_p]),
    "rlx_sac_replay_draw_i32": (c_int, [c_void_p, _U32P, c_int, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "rlx_sac_act_f32": (c_int, [c_void_p, _DESCP, c_void_p, c_void_p, _U32P, c_int, c_void_p, c_int, c_float, c_float,
                                c_int, c_int, c_int, c_void_p]),
    "rlx_sac_act_processed_f32": (c_int, [c_void_p, _DESCP, c_void_p, c_void_p, _U32P, c_int, c_void_p, c_int, c_float, c_float,
                                          c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rlx_sac_update_f32": (c_int, [c_void_p, _DESCP, c_void_p, c_void_p, c_void_p, _DESCP] + [c_void_p] * 12
                           + [c_int64, _U32P, c_int, _I64P, _SACHPP, c_void_p, c_void_p]),
    "rlx_lstm_policy_param_count": (c_int64, [_LDESCP]),
    "rlx_ppo_lstm_rollout_begin": (c_int, [c_void_p, _LDESCP, c_void_p, _DESCP, c_void_p, c_void_p]),
    "rlx_ppo_lstm_act_f32": (c_int, [c_void_p, _LDESCP, c_void_p, _DESCP, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, _U32P, c_int,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int,
                                     c_int, c_void_p]),
    "rlx_lstm_mask_carry_f32": (c_int, [c_void_p] * 6 + [c_int, c_int, c_void_p]),
    "rlx_ppo_lstm_minibatch_fwd_bwd_f32": (c_int, [c_void_p, _LDESCP, c_void_p, c_void_p, _DESCP] + [c_void_p] * 12
                                           + [c_int, c_int, c_int, _HPP, c_void_p]),
    "rlx_ppo_lstm_update_f32": (c_int, [c_void_p, _LDESCP, c_void_p, c_void_p, c_void_p, _DESCP] + [c_void_p] * 11
                                + [c_int, c_int, c_int, c_int, _U32P, c_int, _I64P, _F32HP, _HPP, c_void_p, c_void_p]),
    "rlx_ppo_prefetch_permutation": (c_int, [c_void_p, _U32P, c_int, c_int64, c_int, c_void_p]),
    "rlx_dist_load_rccl": (c_int, [c_char_p]),
    "rlx_dist_unique_id": (c_int, [c_void_p]),
    "rlx_ctx_create_dist": (c_int, [c_int, c_int, c_int, c_void_p, POINTER(c_void_p)]),
    "rlx_ctx_rank": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int)]),
    "rlx_dist_comm_count": (c_int, [c_void_p, POINTER(c_int)]),
    "rlx_allreduce_grads": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "rlx_dist_row_capacity": (c_int, [c_int, c_int, c_int]),
    "rlx_dist_local_rows_i32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                        c_void_p]),
    "rlx_lnmlp_param_count": (c_int64, [POINTER(LnMlpDesc)]),
    "rlx_lnmlp_fwd_f32": (c_int, [c_void_p, POINTER(LnMlpDesc), c_void_p, c_void_p, c_int, c_void_p, c_int64, c_void_p]),
    "rlx_fastsac_replay_sample_f32": (c_int, [c_void_p] * 7 + [c_int] * 5 + [c_float, c_int, c_int, c_void_p, c_void_p, c_int64] + [c_void_p] * 8),
    "rlx_fastsac_act_f32": (c_int, [c_void_p, POINTER(LnMlpDesc), c_void_p, c_void_p, c_void_p, _U32P, c_int, c_void_p, c_int, c_int, c_int,
                                    c_int, POINTER(FastSacHparams), c_void_p]),
    "rlx_fastsac_critic_update_f32": (c_int, [c_void_p, POINTER(LnMlpDesc), c_void_p, POINTER(LnMlpDesc)] + [c_void_p] * 17 +
                                      [c_int64, _U32P, c_int, POINTER(c_int64), POINTER(FastSacHparams), c_void_p, c_void_p]),
    "rlx_fastsac_policy_update_f32": (c_int, [c_void_p, POINTER(LnMlpDesc), c_void_p, c_void_p, c_void_p, POINTER(LnMlpDesc), c_void_p,
                                              c_void_p, c_void_p, c_void_p, c_void_p, c_int64, _U32P, c_int, POINTER(c_int64),
                                              POINTER(FastSacHparams), c_void_p, c_void_p]),
    "rlx_dist_overflow_count": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int), c_void_p]),
    "rlx_ppo_dist_prefetch": (c_int, [c_void_p, _U32P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "rlx_ppo_update_dist_f32": (c_int, [c_void_p, _DESCP, c_void_p, c_void_p, c_void_p, _DESCP, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                        c_int, _U32P, c_int, _I64P, _F32HP, _HPP, c_void_p, c_void_p]),
    "rlx_dbg_set_allreduce_hook": (c_int, [c_void_p, c_void_p, c_void_p]),
    "rlx_dbg_set_rank": (c_int, [c_void_p, c_int, c_int]),
    "rlx_ctx_side_stream": (c_void_p, [c_void_p]),
    "rlx_ppo_update_f32": (c_int, [c_void_p, _DESCP, c_void_p, c_void_p, c_void_p, _DESCP, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                   c_int, _U32P, c_int, _I64P, _F32HP, _HPP, c_void_p, c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)
_lib = None


def load_library():
    """Load librlxhip.so (once).  Raises RlxError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch must be imported BEFORE librlxhip.so is mapped: both depend on libamdhip64.so.7 and the
    # process must use ONE HIP runtime -- torch's bundled one, the one that owns the tensors we are
    # handed (loading /opt/rocm's copy first was observed to leave the GPU undetected on the GPU box).
    import torch  # noqa: F401
    path = library_path()
    if not os.path.exists(path):
        raise RlxError(f"{path} not found: build it with `python rl-x_amd/build.py` "
                       "(there is no CPU fallback for the HIP hot path)")
    lib = ctypes.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _check(rc, what):
    if rc != 0:
        msg = load_library().rlx_last_error()
        raise RlxError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


def _ptr(t, dtype=None, allow_none=False):
    """device pointer of a contiguous CUDA/HIP torch tensor"""
    if t is None:
        if allow_none:
            return None
        raise RlxError("NULL tensor")
    if not t.is_cuda:
        raise RlxError("tensor must live on the GPU (the HIP hot path has no CPU fallback)")
    if not t.is_contiguous():
        raise RlxError("tensor must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise RlxError(f"tensor dtype {t.dtype} != {dtype}")
    return c_void_p(t.data_ptr())


_RAW_STREAM = None


def _stream():
    """torch's current stream on the current device as a raw hipStream_t (the library launches on the caller's stream).
    torch.cuda.current_stream() builds a Stream object through five Python frames (9 us per call, four calls per SAC vector
    step); the raw accessors underneath it cost 0.3 us."""
    global _RAW_STREAM
    if _RAW_STREAM is None:
        import torch
        raw, dev = getattr(torch._C, "_cuda_getCurrentRawStream", None), getattr(torch._C, "_cuda_getDevice", None)
        if raw is not None and dev is not None:
            _RAW_STREAM = lambda: raw(dev())
        else:
            _RAW_STREAM = lambda: torch.cuda.current_stream().cuda_stream
    return c_void_p(_RAW_STREAM())


def _key_arr(key):
    a = (c_uint32 * 2)(int(key[0]), int(key[1]))
    return a


def threefry_split(key, num=2, scheme=THREEFRY_PARTITIONABLE):
    """jax.random.split(key, num) on the host -> np.uint32[num,2]."""
    lib = load_library()
    out = (c_uint32 * (2 * num))()
    _check(lib.rlx_threefry_split_host(_key_arr(key), out, num, scheme), "rlx_threefry_split_host")
    return np.ctypeslib.as_array(out).reshape(num, 2).copy()


def prng_key(seed):
    seed = int(seed)
    return np.array([(seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF], dtype=np.uint32)


def load_rccl():
    """Bind the library to the RCCL PyTorch-ROCm ships (one RCCL per process)."""
    import torch
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    _check(load_library().rlx_dist_load_rccl(cand.encode() if os.path.exists(cand) else None), "rlx_dist_load_rccl")


def rccl_path():
    """What the library's RCCL entry points are bound to ("" before the first bind)."""
    return load_library().rlx_dist_rccl_path().decode()


def nccl_unique_id():
    """rank 0: 128-byte RCCL unique id (bytes) to broadcast to the other ranks."""
    load_rccl()
    buf = ctypes.create_string_buffer(128)
    _check(load_library().rlx_dist_unique_id(buf), "rlx_dist_unique_id")
    return buf.raw


class Ctx:
    """Owns an rlx_ctx (scratch arenas) on one GPU; methods launch on torch's current stream."""

    def __init__(self, device=0, rank=0, world=1, unique_id=None):
        """world > 1: one rank of a data-parallel job -- the context owns an RCCL communicator (unique_id: the 128 bytes
        rank 0 obtained from nccl_unique_id(), broadcast by the host); collective over all ranks."""
        import torch
        self.lib = load_library()
        self.torch = torch
        if not torch.cuda.is_available():
            raise RlxError("no HIP device visible: the rlx_amd hot path runs on MI355X only (no CPU fallback)")
        self.device = int(device)
        h = c_void_p()
        if world > 1 or unique_id is not None:
            load_rccl()
            if unique_id is None or len(bytes(unique_id)) != 128:
                raise RlxError("world > 1 needs the 128-byte RCCL unique id of rank 0")
            idbuf = ctypes.create_string_buffer(bytes(unique_id), 128)
            _check(self.lib.rlx_ctx_create_dist(self.device, int(rank), int(world), idbuf, ctypes.byref(h)),
                   "rlx_ctx_create_dist")
        else:
            _check(self.lib.rlx_ctx_create(self.device, ctypes.byref(h)), "rlx_ctx_create")
        self.h = h
        self._hook = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.rlx_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- live kernel timing (bench.py roofline leg)
    def prof_begin(self):
        _check(self.lib.rlx_prof_begin(self.h), "rlx_prof_begin")

    def prof_end(self):
        """-> {kernel name: (total ms, total algorithmic FLOPs, launches, total algorithmic bytes)} for the MFMA kernels."""
        n = self.lib.rlx_prof_kernel_count()
        ms, fl, by, cnt = (ctypes.c_double * n)(), (ctypes.c_double * n)(), (ctypes.c_double * n)(), (c_int64 * n)()
        _check(self.lib.rlx_prof_end(self.h, ms, fl, by, cnt), "rlx_prof_end")
        return {self.lib.rlx_prof_kernel_name(i).decode(): (ms[i], fl[i], cnt[i], by[i]) for i in range(n)}

    def prof_rows(self):
        """After prof_end: one dict per (kernel kind, engine, GEMM shape): every launch counted, every prof_sample-th launch
        of the row timed (ms / flops / bytes are sums over the timed ones)."""
        n = c_int(0)
        _check(self.lib.rlx_prof_rows(self.h, None, 0, ctypes.byref(n)), "rlx_prof_rows")
        rows = (ProfRow * max(n.value, 1))()
        _check(self.lib.rlx_prof_rows(self.h, rows, n.value, ctypes.byref(n)), "rlx_prof_rows")
        return [dict(kernel=self.lib.rlx_prof_kernel_name(r.kernel).decode(), engine=int(r.engine), M=int(r.M), N=int(r.N),
                     K=int(r.K), launches=int(r.launches), timed=int(r.timed), ms=float(r.ms), flops=float(r.flops),
                     bytes=float(r.bytes)) for r in rows[:n.value]]

    def prof_union_ms(self):
        """After prof_end: ms during which at least one instrumented kernel was running (union over streams)."""
        out = ctypes.c_double()
        _check(self.lib.rlx_prof_union_ms(self.h, ctypes.byref(out)), "rlx_prof_union_ms")
        return out.value

    def set_option(self, name, value):
        _check(self.lib.rlx_dbg_set_option(self.h, name.encode(), int(value)), "rlx_dbg_set_option")

    def get_counter(self, name):
        out = c_int64()
        _check(self.lib.rlx_dbg_get_counter(self.h, name.encode(), ctypes.byref(out)), "rlx_dbg_get_counter")
        return out.value

    def dbg_set_sac_noise(self, eps_next=None, eps_cur=None):
        f = self.torch.float32
        _check(self.lib.rlx_dbg_set_sac_noise(self.h, _ptr(eps_next, f, True), _ptr(eps_cur, f, True)), "rlx_dbg_set_sac_noise")

    def dbg_set_stamps(self, stamps=None):
        """stamps: device int64 tensor of >= 8 elements (phase clock stamps of instrumented kernels), or None."""
        _check(self.lib.rlx_dbg_set_stamps(self.h, c_void_p(stamps.data_ptr()) if stamps is not None else None), "rlx_dbg_set_stamps")

    def dbg_gemm(self, mode, A, B, C, aux, M, N, K, act):
        f = self.torch.float32
        _check(self.lib.rlx_dbg_gemm_f32(self.h, mode, _ptr(A, f), _ptr(B, f), _ptr(C, f), _ptr(aux, f, True), M, N, K,
                                         act, _stream()), "rlx_dbg_gemm_f32")

    def dbg_l1(self, bwd, X, W, b, g, be, H, ln_partials, act, ln, grid):
        f = self.torch.float32
        M, O = X.shape
        _check(self.lib.rlx_dbg_l1_f32(self.h, int(bwd), _ptr(X, f), _ptr(W, f), _ptr(b, f), _ptr(g, f, True),
                                       _ptr(be, f, True), _ptr(H, f), _ptr(ln_partials, f, True), M, O, H.shape[1],
                                       act, int(ln), grid, _stream()), "rlx_dbg_l1_f32")

    # ---- PRNG
    def random_bits(self, key, out, scheme=THREEFRY_PARTITIONABLE):
        t = self.torch
        _check(self.lib.rlx_random_bits_u32(self.h, _key_arr(key), _ptr(out, t.int32), out.numel(), scheme, _stream()),
               "rlx_random_bits_u32")

    def normal(self, key, out, scheme=THREEFRY_PARTITIONABLE):
        t = self.torch
        _check(self.lib.rlx_normal_f32(self.h, _key_arr(key), _ptr(out, t.float32), out.numel(), scheme, _stream()),
               "rlx_normal_f32")

    def permutation(self, key, out, E, B, scheme=THREEFRY_PARTITIONABLE):
        """Advances and returns the key; fills out (int32[E*B])."""
        t = self.torch
        k = _key_arr(key)
        if out.numel() != E * B:
            raise RlxError("permutation: out must hold E*B int32")
        _check(self.lib.rlx_permutation_i32(self.h, k, _ptr(out, t.int32), E, B, scheme, _stream()),
               "rlx_permutation_i32")
        return np.array([k[0], k[1]], dtype=np.uint32)

    # ---- env
    def env_reset(self, seed, env_id_offset, horizon, obs, ep_step, ep_ret, last_ret, last_len):
        t = self.torch
        N, O = obs.shape
        _check(self.lib.rlx_env_reset_f32(self.h, seed, env_id_offset, N, O, horizon, _ptr(obs, t.float32),
                                          _ptr(ep_step, t.int32), _ptr(ep_ret, t.float32), _ptr(last_ret, t.float32),
                                          _ptr(last_len, t.float32), _stream()), "rlx_env_reset_f32")

    def env_step(self, seed, env_id_offset, t_step, horizon, p_term, reward_noise, action, obs, final_obs, reward,
                 terminated, truncated, ep_step, ep_ret, last_ret, last_len, episode_stats=None, prev_obs_out=None):
        """prev_obs_out (optional, [N, O]): receives the PRE-step observation (the replay ring's `states` row)."""
        t = self.torch
        N, O = obs.shape
        A = action.shape[1]
        f = t.float32
        _check(self.lib.rlx_env_step_copy_f32(self.h, seed, env_id_offset, int(t_step) & 0xFFFFFFFF, N, O, A, horizon,
                                              p_term, reward_noise, _ptr(action, f), _ptr(obs, f), _ptr(final_obs, f),
                                              _ptr(reward, f), _ptr(terminated, f), _ptr(truncated, f),
                                              _ptr(ep_step, t.int32), _ptr(ep_ret, f), _ptr(last_ret, f), _ptr(last_len, f),
                                              _ptr(episode_stats, f, True), _ptr(prev_obs_out, f, True), _stream()),
               "rlx_env_step_copy_f32")

    # ---- acting
    def actor_critic_fwd_sample(self, pdesc, pparams, cdesc, cparams, obs, key, action, processed, value, logp,
                                states_row=None, clip_and_rescale=False, act_low=None, act_high=None,
                                scheme=THREEFRY_PARTITIONABLE, env_id_offset=0, n_global=None, critic_obs=None):
        """Advances and returns the key.  critic_obs: the critic's own observation columns [N, cdesc.in_dim] (None: obs)."""
        f = self.torch.float32
        k = _key_arr(key)
        _check(self.lib.rlx_actor_critic_fwd_sample_f32(
            self.h, ctypes.byref(pdesc), _ptr(pparams, f), ctypes.byref(cdesc), _ptr(cparams, f), _ptr(obs, f),
            _ptr(critic_obs, f, True), k,
            scheme, _ptr(action, f), _ptr(processed, f, True), _ptr(value, f), _ptr(logp, f),
            _ptr(states_row, f, True), obs.shape[0], int(bool(clip_and_rescale)), _ptr(act_low, f, True),
            _ptr(act_high, f, True), int(env_id_offset), int(n_global or obs.shape[0]), _stream()),
            "rlx_actor_critic_fwd_sample_f32")
        return np.array([k[0], k[1]], dtype=np.uint32)

    def c51_critic_loss(self, q1, q2, q1_next, q2_next, rewards, dones, truncations, n_steps, next_log_probs, log_alpha, gamma,
                        v_min, v_max, clipped_double_q, d_q1, d_q2, out4):
        """FastSAC's distributional critic step (include/rlx_hip.h): logits [B, nr_atoms]; out4 = {q_loss, q_min, q_max, 0}."""
        f = self.torch.float32
        B, NA = q1.shape
        _check(self.lib.rlx_c51_critic_loss_f32(self.h, _ptr(q1, f), _ptr(q2, f), _ptr(q1_next, f), _ptr(q2_next, f), _ptr(rewards, f),
                                                _ptr(dones, f), _ptr(truncations, f), _ptr(n_steps, f), _ptr(next_log_probs, f),
                                                _ptr(log_alpha, f), int(B), int(NA), gamma, v_min, v_max, int(bool(clipped_double_q)),
                                                _ptr(d_q1, f), _ptr(d_q2, f), _ptr(out4, f), _stream()), "rlx_c51_critic_loss_f32")
        return out4

    def obs_norm_update(self, obs, mean, var, std, count):
        """FastSAC's running observation statistics (include/rlx_hip.h): obs [B, O]; mean / var / std fp32 [O], count int64[1]."""
        t = self.torch
        f = t.float32
        _check(self.lib.rlx_obs_norm_update_f32(self.h, _ptr(obs, f), int(obs.shape[0]), int(obs.shape[1]), _ptr(mean, f),
                                                _ptr(var, f), _ptr(std, f), _ptr(count, t.int64), _stream()), "rlx_obs_norm_update_f32")

    def obs_norm_apply(self, obs, mean, std, out, eps=1e-8):
        f = self.torch.float32
        _check(self.lib.rlx_obs_norm_apply_f32(self.h, _ptr(obs, f), int(obs.shape[0]), int(obs.shape[1]), _ptr(mean, f),
                                               _ptr(std, f), eps, _ptr(out, f), _stream()), "rlx_obs_norm_apply_f32")
        return out

    def select_columns(self, x, cols, out):
        """out[m, j] = x[m, cols[j]]  (x [M, ldx], cols int32 [n] on the device, out [M, >= n])."""
        t = self.torch
        M = int(x.numel() // x.shape[-1])
        _check(self.lib.rlx_select_columns_f32(self.h, _ptr(x, t.float32), int(x.shape[-1]), _ptr(cols, t.int32), int(cols.numel()),
                                               _ptr(out, t.float32), int(out.shape[-1]), M, _stream()), "rlx_select_columns_f32")
        return out

    def actor_critic_fwd_sample_discrete(self, pdesc, pparams, cdesc, cparams, obs, key, action, value, logp, states_row=None,
                                         scheme=THREEFRY_PARTITIONABLE, env_id_offset=0, n_global=None, deterministic=False):
        """Categorical policy: action [N] = sampled index (float), logp [N], value [N].  Returns the new key."""
        f = self.torch.float32
        k = _key_arr(key)
        _check(self.lib.rlx_actor_critic_fwd_sample_discrete_f32(
            self.h, ctypes.byref(pdesc), _ptr(pparams, f), ctypes.byref(cdesc), _ptr(cparams, f), _ptr(obs, f), k, scheme,
            _ptr(action, f), _ptr(value, f), _ptr(logp, f), _ptr(states_row, f, True), obs.shape[0], int(env_id_offset),
            int(n_global or obs.shape[0]), int(bool(deterministic)), _stream()), "rlx_actor_critic_fwd_sample_discrete_f32")
        return np.array([k[0], k[1]], dtype=np.uint32)

    def rollout_step_supported(self, pdesc, cdesc):
        return bool(self.lib.rlx_ppo_rollout_step_supported(ctypes.byref(pdesc), ctypes.byref(cdesc)))

    def rollout_begin(self, pdesc, pparams, cdesc, cparams):
        """Weight images for the acting steps that follow (valid until the parameters change; see include/rlx_hip.h)."""
        f = self.torch.float32
        _check(self.lib.rlx_ppo_rollout_begin(self.h, ctypes.byref(pdesc), _ptr(pparams, f), ctypes.byref(cdesc),
                                              _ptr(cparams, f), _stream()), "rlx_ppo_rollout_begin")

    def rollout_end(self):
        _check(self.lib.rlx_ppo_rollout_end(self.h), "rlx_ppo_rollout_end")

    def rollout_step(self, pdesc, pparams, cdesc, cparams, obs_in, obs_out, key, action, processed, value, logp,
                     clip_and_rescale=False, act_low=None, act_high=None, scheme=THREEFRY_PARTITIONABLE,
                     noise_row_offset=0, n_global=None, env=None):
        """Fused acting step.  env: None, or a dict with the synthetic env's state (see RandomObsEnv.fused_args)."""
        t = self.torch
        f = t.float32
        k = _key_arr(key)
        N = obs_in.shape[0]
        e = env or {}
        _check(self.lib.rlx_ppo_rollout_step_f32(
            self.h, ctypes.byref(pdesc), _ptr(pparams, f), ctypes.byref(cdesc), _ptr(cparams, f), _ptr(obs_in, f),
            _ptr(obs_out, f, True), k, scheme, _ptr(action, f), _ptr(processed, f, True), _ptr(value, f), _ptr(logp, f),
            N, int(bool(clip_and_rescale)), _ptr(act_low, f, True), _ptr(act_high, f, True), int(noise_row_offset),
            int(n_global or N), 1 if env else 0, int(e.get("seed", 0)), int(e.get("env_id_offset", 0)),
            int(e.get("t", 0)) & 0xFFFFFFFF, int(e.get("horizon", 1)), float(e.get("p_term", 0.0)),
            float(e.get("reward_noise", 0.0)), _ptr(e.get("final_obs"), f, True), _ptr(e.get("reward"), f, True),
            _ptr(e.get("terminated"), f, True), _ptr(e.get("ep_step"), t.int32, True), _ptr(e.get("ep_ret"), f, True),
            _ptr(e.get("last_ret"), f, True), _ptr(e.get("last_len"), f, True), _ptr(e.get("episode_stats"), f, True),
            _stream()), "rlx_ppo_rollout_step_f32")
        return np.array([k[0], k[1]], dtype=np.uint32)

    def rollout(self, pdesc, pparams, cdesc, cparams, states, obs_last, key, actions, values, logps, env,
                clip_and_rescale=False, act_low=None, act_high=None, scheme=THREEFRY_PARTITIONABLE, noise_row_offset=0,
                n_global=None):
        """The T fused acting steps of one rollout, queued by one library call (include/rlx_hip.h: rlx_ppo_rollout_f32).
        states [T,N,O] (row 0 = the current observation), actions [T,N,A], values / logps [T,N]; env: the synthetic env's
        state as RandomObsEnv.fused_args gives it, with final_obs [T,N,O], reward / terminated [T,N]."""
        t = self.torch
        f = t.float32
        k = _key_arr(key)
        T, N = values.shape
        for x in (states, actions, values, logps, env["final_obs"], env["reward"], env["terminated"]):
            assert x.is_contiguous() and x.shape[0] == T
        _check(self.lib.rlx_ppo_rollout_f32(
            self.h, ctypes.byref(pdesc), _ptr(pparams, f), ctypes.byref(cdesc), _ptr(cparams, f), _ptr(states, f),
            _ptr(obs_last, f), k, scheme, _ptr(actions, f), _ptr(values, f), _ptr(logps, f), T, N,
            int(bool(clip_and_rescale)), _ptr(act_low, f, True), _ptr(act_high, f, True), int(noise_row_offset),
            int(n_global or N), int(env["seed"]), int(env["env_id_offset"]), int(env["t"]) & 0xFFFFFFFF, int(env["horizon"]),
            float(env["p_term"]), float(env["reward_noise"]), _ptr(env["final_obs"], f), _ptr(env["reward"], f),
            _ptr(env["terminated"], f), _ptr(env["ep_step"], t.int32), _ptr(env["ep_ret"], f), _ptr(env["last_ret"], f),
            _ptr(env["last_len"], f), _ptr(env.get("episode_stats"), f, True), _stream()), "rlx_ppo_rollout_f32")
        return np.array([k[0], k[1]], dtype=np.uint32)

    def mlp_fwd(self, desc, params, x, out):
        f = self.torch.float32
        n = x.numel() // desc.in_dim
        _check(self.lib.rlx_mlp_fwd_f32(self.h, ctypes.byref(desc), _ptr(params, f), _ptr(x, f), _ptr(out, f), n,
                                        _stream()), "rlx_mlp_fwd_f32")

    # ---- GAE
    def ppo_next_values(self, cdesc, cparams, states, next_states, values, next_values):
        """next_values[t] = critic(next_states[t]), reusing values[t+1] where next_states[t] == states[t+1] (device-side)."""
        f = self.torch.float32
        T, N = values.shape
        _check(self.lib.rlx_ppo_next_values_f32(self.h, ctypes.byref(cdesc), _ptr(cparams, f), _ptr(states, f),
                                                _ptr(next_states, f), _ptr(values, f), _ptr(next_values, f), T, N, _stream()),
               "rlx_ppo_next_values_f32")

    def gae(self, rewards, values, next_values, terminations, advantages, returns, gamma, gae_lambda):
        f = self.torch.float32
        T, N = rewards.shape
        _check(self.lib.rlx_gae_f32(self.h, _ptr(rewards, f), _ptr(values, f), _ptr(next_values, f),
                                    _ptr(terminations, f), _ptr(advantages, f), _ptr(returns, f), T, N, gamma,
                                    gae_lambda, _stream()), "rlx_gae_f32")

    def ppo_reduce_metrics(self, metrics, returns, values, logstd, out12):
        """out12 (device, 12 floats) <- means of the [n_upd, 10] metric rows, explained variance, mean policy std."""
        f = self.torch.float32
        _check(self.lib.rlx_ppo_reduce_metrics_f32(self.h, _ptr(metrics, f), int(metrics.shape[0]), _ptr(returns, f), _ptr(values, f),
                                                   int(returns.numel()), _ptr(logstd, f) if logstd is not None else None,
                                                   int(logstd.numel()) if logstd is not None else 0, _ptr(out12, f), _stream()),
               "rlx_ppo_reduce_metrics_f32")
        return out12

    # ---- minibatch loss + grads
    def ppo_minibatch_fwd_bwd(self, pdesc, pparams, pgrads, cdesc, cparams, cgrads, metrics, states, actions,
                              log_probs, returns, advantages, idx, hp, mb_global=None, stats_io=None, phase=1):
        t = self.torch
        f = t.float32
        mb_local = idx.numel()
        _check(self.lib.rlx_ppo_minibatch_fwd_bwd_f32(
            self.h, ctypes.byref(pdesc), _ptr(pparams, f), _ptr(pgrads, f, True), ctypes.byref(cdesc), _ptr(cparams, f),
            _ptr(cgrads, f, True), _ptr(metrics, f), _ptr(states, f), _ptr(actions, f), _ptr(log_probs, f), _ptr(returns, f),
            _ptr(advantages, f), _ptr(idx, t.int32), mb_local, mb_global or mb_local, _ptr(stats_io, t.float64, True), phase,
            ctypes.byref(hp), _stream()), "rlx_ppo_minibatch_fwd_bwd_f32")

    # ---- optimizer
    def grad_global_norm(self, grads, norm_out):
        f = self.torch.float32
        _check(self.lib.rlx_grad_global_norm_f32(self.h, _ptr(grads, f), grads.numel(), _ptr(norm_out, f), _stream()),
               "rlx_grad_global_norm_f32")

    def clip_adam_step(self, params, grads, m, v, step, lr, max_grad_norm, b1=0.9, b2=0.999, eps=1e-8,
                       grad_norm_out=None):
        f = self.torch.float32
        _check(self.lib.rlx_clip_adam_step_f32(self.h, _ptr(params, f), _ptr(grads, f), _ptr(m, f), _ptr(v, f),
                                               params.numel(), step, lr, max_grad_norm, b1, b2, eps,
                                               _ptr(grad_norm_out, f, True), _stream()), "rlx_clip_adam_step_f32")

    # ---- SAC
    def sac_replay_sample(self, ring, idx1, idx2, out):
        """ring / out: tuples (states, next_states, actions, rewards, terminations)."""
        t = self.torch
        f = t.float32
        cap, N, O = ring[0].shape
        A = ring[2].shape[2]
        B = idx1.numel()
        _check(self.lib.rlx_sac_replay_sample_f32(
            self.h, *[_ptr(x, f) for x in ring], N, O, A, _ptr(idx1, t.int32), _ptr(idx2, t.int32), B,
            *[_ptr(x, f) for x in out], _stream()), "rlx_sac_replay_sample_f32")

    def sac_invalidate_images(self):
        """after writing SAC parameter / target vectors from outside rlx_sac_update_f32 (rlx_sac_hparams.keep_images contract)"""
        _check(self.lib.rlx_sac_invalidate_images(self.h), "rlx_sac_invalidate_images")

    def sac_replay_draw(self, update_key, B, size, nr_envs, idx1, idx2, scheme=THREEFRY_PARTITIONABLE):
        """Device-side index draw of the fully jitted SAC flavour (same key for both index vectors)."""
        t = self.torch
        _check(self.lib.rlx_sac_replay_draw_i32(self.h, _key_arr(update_key), scheme, int(B), int(size), int(nr_envs),
                                                _ptr(idx1, t.int32), _ptr(idx2, t.int32), _stream()), "rlx_sac_replay_draw_i32")

    def sac_act(self, pdesc, pparams, obs, key, action, log_std_min, log_std_max, deterministic=False,
                scheme=THREEFRY_PARTITIONABLE, row_offset=0, n_global=None, processed=None):
        """processed = (low [A], half_range [A], out [N, A]): the env-facing action of get_processed_action from the same launch."""
        f = self.torch.float32
        k = _key_arr(key)
        N = obs.shape[0]
        if processed is not None:
            low, half, out = processed
            _check(self.lib.rlx_sac_act_processed_f32(self.h, ctypes.byref(pdesc), _ptr(pparams, f), _ptr(obs, f), k, scheme,
                                                      _ptr(action, f), N, log_std_min, log_std_max, int(bool(deterministic)),
                                                      int(row_offset), int(n_global or N), _ptr(low, f), _ptr(half, f),
                                                      _ptr(out, f), _stream()), "rlx_sac_act_processed_f32")
            return np.array([k[0], k[1]], dtype=np.uint32)
        _check(self.lib.rlx_sac_act_f32(self.h, ctypes.byref(pdesc), _ptr(pparams, f), _ptr(obs, f), k, scheme,
                                        _ptr(action, f), N, log_std_min, log_std_max, int(bool(deterministic)),
                                        int(row_offset), int(n_global or N), _stream()), "rlx_sac_act_f32")
        return np.array([k[0], k[1]], dtype=np.uint32)

    def sac_update(self, pdesc, pparams, pm, pv, qdesc, qparams, qm, qv, qtarget, log_alpha, am, av, batch, key,
                   opt_count, hp, metrics_out, scheme=THREEFRY_PARTITIONABLE):
        """batch = (states, next_states, actions, rewards, terminations).  Returns (new_key, new_opt_count).
        With the ring source (hp.ring_*) states and next_states may both be None: the gathered observation rows are not written."""
        f = self.torch.float32
        k = _key_arr(key)
        cnt = c_int64(int(opt_count))
        B = batch[2].shape[0]
        _check(self.lib.rlx_sac_update_f32(
            self.h, ctypes.byref(pdesc), _ptr(pparams, f), _ptr(pm, f), _ptr(pv, f), ctypes.byref(qdesc),
            _ptr(qparams, f), _ptr(qm, f), _ptr(qv, f), _ptr(qtarget, f), _ptr(log_alpha, f), _ptr(am, f), _ptr(av, f),
            *[_ptr(x, f, allow_none=i < 2) for i, x in enumerate(batch)], B, k, scheme, ctypes.byref(cnt), ctypes.byref(hp),
            _ptr(metrics_out, f), _stream()), "rlx_sac_update_f32")
        return np.array([k[0], k[1]], dtype=np.uint32), cnt.value

    # ---- PPO + LSTM
    def lstm_policy_param_count(self, desc):
        return int(self.lib.rlx_lstm_policy_param_count(ctypes.byref(desc)))

    def ppo_lstm_rollout_begin(self, desc, pparams, cdesc, cparams):
        f = self.torch.float32
        _check(self.lib.rlx_ppo_lstm_rollout_begin(self.h, ctypes.byref(desc), _ptr(pparams, f), ctypes.byref(cdesc),
                                                   _ptr(cparams, f), _stream()), "rlx_ppo_lstm_rollout_begin")

    def ppo_lstm_act(self, desc, pparams, cdesc, cparams, obs, c, h, key, action, processed, value, logp,
                     clip_and_rescale=False, act_low=None, act_high=None, scheme=THREEFRY_PARTITIONABLE,
                     noise_row_offset=0, n_global=None, deterministic=False, critic_obs=None):
        """Policy.apply_one_step + sampling + critic value; carry (c, h) updated in place.  Returns the new key.
        critic_obs: [N, cdesc.in_dim], the critic's own observation columns (None: it reads `obs`)."""
        f = self.torch.float32
        k = _key_arr(key)
        N = obs.shape[0]
        _check(self.lib.rlx_ppo_lstm_act_f32(
            self.h, ctypes.byref(desc), _ptr(pparams, f), ctypes.byref(cdesc), _ptr(cparams, f), _ptr(obs, f),
            _ptr(critic_obs, f, True), _ptr(c, f),
            _ptr(h, f), k, scheme, _ptr(action, f), _ptr(processed, f, True), _ptr(value, f), _ptr(logp, f), N,
            int(bool(clip_and_rescale)), _ptr(act_low, f, True), _ptr(act_high, f, True), int(noise_row_offset),
            int(n_global or N), int(bool(deterministic)), _stream()), "rlx_ppo_lstm_act_f32")
        return np.array([k[0], k[1]], dtype=np.uint32)

    def lstm_mask_carry(self, c, h, terminated, truncated=None, done_out=None):
        f = self.torch.float32
        _check(self.lib.rlx_lstm_mask_carry_f32(self.h, _ptr(c, f), _ptr(h, f), _ptr(terminated, f), _ptr(truncated, f, True),
                                                _ptr(done_out, f, True), c.shape[0], c.shape[1], _stream()),
               "rlx_lstm_mask_carry_f32")

    def ppo_lstm_minibatch_fwd_bwd(self, desc, pparams, pgrads, cdesc, cparams, cgrads, metrics, states, actions, log_probs,
                                   returns, advantages, dones, c0, h0, env_idx, hp):
        t = self.torch
        f = t.float32
        T, N = log_probs.shape
        _check(self.lib.rlx_ppo_lstm_minibatch_fwd_bwd_f32(
            self.h, ctypes.byref(desc), _ptr(pparams, f), _ptr(pgrads, f), ctypes.byref(cdesc), _ptr(cparams, f),
            _ptr(cgrads, f), _ptr(metrics, f), _ptr(states, f), _ptr(actions, f), _ptr(log_probs, f), _ptr(returns, f),
            _ptr(advantages, f), _ptr(dones, f), _ptr(c0, f), _ptr(h0, f), _ptr(env_idx, t.int32), env_idx.numel(), T, N,
            ctypes.byref(hp), _stream()), "rlx_ppo_lstm_minibatch_fwd_bwd_f32")

    def ppo_lstm_update(self, desc, pparams, pm, pv, cdesc, cparams, cm, cv, states, actions, log_probs, returns,
                        advantages, dones, c0, h0, nr_epochs, minibatch_size, key, opt_count, lr_schedule, hp, metrics_out,
                        scheme=THREEFRY_PARTITIONABLE):
        """Returns (new_key, new_opt_count)."""
        f = self.torch.float32
        T, N = log_probs.shape
        k = _key_arr(key)
        cnt = c_int64(int(opt_count))
        lr = np.ascontiguousarray(lr_schedule, dtype=np.float32)
        _check(self.lib.rlx_ppo_lstm_update_f32(
            self.h, ctypes.byref(desc), _ptr(pparams, f), _ptr(pm, f), _ptr(pv, f), ctypes.byref(cdesc), _ptr(cparams, f),
            _ptr(cm, f), _ptr(cv, f), _ptr(states, f), _ptr(actions, f), _ptr(log_probs, f), _ptr(returns, f),
            _ptr(advantages, f), _ptr(dones, f), _ptr(c0, f), _ptr(h0, f), T, N, nr_epochs, minibatch_size, k, scheme,
            ctypes.byref(cnt), lr.ctypes.data_as(_F32HP), ctypes.byref(hp), _ptr(metrics_out, f), _stream()),
            "rlx_ppo_lstm_update_f32")
        return np.array([k[0], k[1]], dtype=np.uint32), cnt.value

    def ppo_prefetch_permutation(self, key_at_update, nr_epochs, batch_size, scheme=THREEFRY_PARTITIONABLE):
        """Generate the next update's permutation on the library's side stream, ordered after the current torch stream."""
        k = _key_arr(key_at_update)
        _check(self.lib.rlx_ppo_prefetch_permutation(self.h, k, nr_epochs, batch_size, scheme, _stream()),
               "rlx_ppo_prefetch_permutation")

    # ---- whole update
    def side_stream(self):
        """The library-owned side stream as a torch stream (the critic chain of the fused updates runs on it)."""
        if getattr(self, "_side_stream", None) is None:
            ptr = self.lib.rlx_ctx_side_stream(self.h)
            if not ptr:
                raise RlxError("rlx_ctx_side_stream failed")
            self._side_stream = self.torch.cuda.ExternalStream(ptr, device=self.torch.device("cuda", self.device))
        return self._side_stream

    # ---- data-parallel update (dist.hip / rlx_ppo_update_dist_f32)
    def rank_world(self):
        r, w = c_int(), c_int()
        _check(self.lib.rlx_ctx_rank(self.h, ctypes.byref(r), ctypes.byref(w)), "rlx_ctx_rank")
        return r.value, w.value

    def comm_count(self):
        """ranks of the context's RCCL communicator (ncclCommCount); 0 without a communicator."""
        n = c_int()
        _check(self.lib.rlx_dist_comm_count(self.h, ctypes.byref(n)), "rlx_dist_comm_count")
        return n.value

    def set_rank(self, rank, world):
        """test hook: rank / world of a context without communicator (emulated ranks)."""
        _check(self.lib.rlx_dbg_set_rank(self.h, int(rank), int(world)), "rlx_dbg_set_rank")

    def set_allreduce_hook(self, fn):
        """test hook: fn(buf_ptr, n, dtype, on_side_stream) stands in for the RCCL collectives (None restores them).
        buf_ptr is the raw device address; dtype 0 = float32, 1 = float64."""
        self._hook_err = []
        if fn is None:
            self._hook = None
            _check(self.lib.rlx_dbg_set_allreduce_hook(self.h, None, None), "rlx_dbg_set_allreduce_hook")
            return

        def _cb(user, buf, n, dtype, on_side):
            try:
                fn(int(buf), int(n), int(dtype), int(on_side))
                return 0
            except BaseException as e:          # never unwind through the C frames
                self._hook_err.append(e)
                return 1
        self._hook = _ALLREDUCE_FN(_cb)          # keep the trampoline alive
        _check(self.lib.rlx_dbg_set_allreduce_hook(self.h, ctypes.cast(self._hook, c_void_p), None),
               "rlx_dbg_set_allreduce_hook")

    def allreduce_grads(self, buf):
        _check(self.lib.rlx_allreduce_grads(self.h, _ptr(buf, self.torch.float32), buf.numel(), _stream()), "rlx_allreduce_grads")

    def dist_row_capacity(self, mb_global, n_local, n_global):
        return int(self.lib.rlx_dist_row_capacity(int(mb_global), int(n_local), int(n_global)))

    def dist_local_rows(self, perm, n_minibatches, mb_global, n_local, n_global, env_id_offset, cap, lidx, counts):
        t = self.torch
        _check(self.lib.rlx_dist_local_rows_i32(self.h, _ptr(perm, t.int32), n_minibatches, mb_global, n_local, n_global,
                                                env_id_offset, cap, _ptr(lidx, t.int32), _ptr(counts, t.int32), _stream()),
               "rlx_dist_local_rows_i32")

    # ---- FastSAC (rl_x/algorithms/fastsac/pytorch)
    def lnmlp_param_count(self, desc):
        return int(self.lib.rlx_lnmlp_param_count(ctypes.byref(desc)))

    def lnmlp_fwd(self, desc, params, x, out):
        f = self.torch.float32
        _check(self.lib.rlx_lnmlp_fwd_f32(self.h, ctypes.byref(desc), _ptr(params, f), _ptr(x, f), int(x.shape[1]), _ptr(out, f),
                                          int(x.shape[0]), _stream()), "rlx_lnmlp_fwd_f32")
        return out

    def fastsac_replay_sample(self, ring, n_steps, gamma, pos, size, idx_t, idx_e, out):
        """ring = (states, next_states, actions, rewards, dones, truncations) [capacity, nr_envs, .]; out = the seven [B, .] outputs"""
        t = self.torch
        f = t.float32
        cap, ne = int(ring[0].shape[0]), int(ring[0].shape[1])
        _check(self.lib.rlx_fastsac_replay_sample_f32(
            self.h, *[_ptr(x, f) for x in ring], cap, ne, int(ring[0].shape[2]), int(ring[2].shape[2]), int(n_steps), float(gamma), int(pos),
            int(size), _ptr(idx_t, t.int32), _ptr(idx_e, t.int32), int(idx_t.numel()), *[_ptr(x, f) for x in out], _stream()),
            "rlx_fastsac_replay_sample_f32")

    def fastsac_act(self, pdesc, pparams, obs, action_scale, key, action, hp, deterministic=False, scheme=THREEFRY_PARTITIONABLE,
                    row_offset=0, n_global=None):
        """policy.get_action; returns the new key"""
        f = self.torch.float32
        k = _key_arr(key)
        N = obs.shape[0]
        _check(self.lib.rlx_fastsac_act_f32(self.h, ctypes.byref(pdesc), _ptr(pparams, f), _ptr(obs, f), _ptr(action_scale, f), k, scheme,
                                            _ptr(action, f), N, int(bool(deterministic)), int(row_offset), int(n_global or N),
                                            ctypes.byref(hp), _stream()), "rlx_fastsac_act_f32")
        return np.array([k[0], k[1]], dtype=np.uint32)

    def fastsac_critic_update(self, pdesc, pparams, qdesc, qparams, qm, qv, qtarget, log_alpha, am, av, batch, action_scale, key,
                              opt_count, hp, metrics_out, scheme=THREEFRY_PARTITIONABLE, critic_states=None, critic_next_states=None):
        """batch = (states, next_states, actions, rewards, dones, truncations, effective_n_steps).  -> (new key, new optimizer count)"""
        f = self.torch.float32
        k = _key_arr(key)
        cnt = c_int64(int(opt_count))
        s, s2, a, r, d, tr, ns = batch
        _check(self.lib.rlx_fastsac_critic_update_f32(
            self.h, ctypes.byref(pdesc), _ptr(pparams, f), ctypes.byref(qdesc), _ptr(qparams, f), _ptr(qm, f), _ptr(qv, f), _ptr(qtarget, f),
            _ptr(log_alpha, f), _ptr(am, f), _ptr(av, f), _ptr(s, f), _ptr(s2, f), _ptr(critic_states, f, True),
            _ptr(critic_next_states, f, True), _ptr(a, f), _ptr(r, f), _ptr(d, f), _ptr(tr, f), _ptr(ns, f), _ptr(action_scale, f),
            int(s.shape[0]), k, scheme, ctypes.byref(cnt), ctypes.byref(hp), _ptr(metrics_out, f), _stream()), "rlx_fastsac_critic_update_f32")
        return np.array([k[0], k[1]], dtype=np.uint32), cnt.value

    def fastsac_policy_update(self, pdesc, pparams, pm, pv, qdesc, qparams, log_alpha, states, action_scale, key, opt_count, hp,
                              metrics_out, scheme=THREEFRY_PARTITIONABLE, critic_states=None):
        f = self.torch.float32
        k = _key_arr(key)
        cnt = c_int64(int(opt_count))
        _check(self.lib.rlx_fastsac_policy_update_f32(
            self.h, ctypes.byref(pdesc), _ptr(pparams, f), _ptr(pm, f), _ptr(pv, f), ctypes.byref(qdesc), _ptr(qparams, f),
            _ptr(log_alpha, f), _ptr(states, f), _ptr(critic_states, f, True), _ptr(action_scale, f), int(states.shape[0]), k, scheme,
            ctypes.byref(cnt), ctypes.byref(hp), _ptr(metrics_out, f), _stream()), "rlx_fastsac_policy_update_f32")
        return np.array([k[0], k[1]], dtype=np.uint32), cnt.value

    def dist_overflow_counts(self):
        """(rows dropped by ANY rank -- identical on every rank, minibatches THIS rank truncated); blocking on the current
        stream, reading resets both"""
        rows, local = c_int(), c_int()
        _check(self.lib.rlx_dist_overflow_count(self.h, ctypes.byref(rows), ctypes.byref(local), _stream()),
               "rlx_dist_overflow_count")
        return rows.value, local.value

    def dist_overflow_count(self):
        """rows dropped by any rank since the last call: the figure every rank agrees on"""
        return self.dist_overflow_counts()[0]

    def ppo_dist_prefetch(self, key_at_update, nr_epochs, T, n_local, n_global, env_id_offset, minibatch_size,
                          scheme=THREEFRY_PARTITIONABLE):
        _check(self.lib.rlx_ppo_dist_prefetch(self.h, _key_arr(key_at_update), nr_epochs, T, n_local, n_global, env_id_offset,
                                              minibatch_size, scheme, _stream()), "rlx_ppo_dist_prefetch")

    def ppo_update_dist(self, pdesc, pparams, pm, pv, cdesc, cparams, cm, cv, states, actions, log_probs, returns,
                        advantages, n_global, env_id_offset, nr_epochs, minibatch_size, key, opt_count, lr_schedule, hp,
                        metrics_out, scheme=THREEFRY_PARTITIONABLE):
        """One rank's share of the data-parallel update; rollout arrays are the local shard [T, N_local, .], minibatch_size
        is global.  Returns (new_key, new_opt_count)."""
        f = self.torch.float32
        T, n_local = log_probs.shape
        k = _key_arr(key)
        cnt = c_int64(int(opt_count))
        lr = np.ascontiguousarray(lr_schedule, dtype=np.float32)
        rc = self.lib.rlx_ppo_update_dist_f32(
            self.h, ctypes.byref(pdesc), _ptr(pparams, f), _ptr(pm, f), _ptr(pv, f), ctypes.byref(cdesc), _ptr(cparams, f),
            _ptr(cm, f), _ptr(cv, f), _ptr(states, f), _ptr(actions, f), _ptr(log_probs, f), _ptr(returns, f),
            _ptr(advantages, f), T, n_local, int(n_global), int(env_id_offset), nr_epochs, minibatch_size, k, scheme,
            ctypes.byref(cnt), lr.ctypes.data_as(_F32HP), ctypes.byref(hp), _ptr(metrics_out, f), _stream())
        if getattr(self, "_hook_err", None):
            e = self._hook_err.pop()
            raise e
        _check(rc, "rlx_ppo_update_dist_f32")
        return np.array([k[0], k[1]], dtype=np.uint32), cnt.value

    def ppo_update(self, pdesc, pparams, pm, pv, cdesc, cparams, cm, cv, states, actions, log_probs, returns,
                   advantages, nr_epochs, minibatch_size, key, opt_count, lr_schedule, hp, metrics_out,
                   scheme=THREEFRY_PARTITIONABLE):
        """Returns (new_key, new_opt_count)."""
        f = self.torch.float32
        T, N = log_probs.shape
        k = _key_arr(key)
        cnt = c_int64(int(opt_count))
        lr = np.ascontiguousarray(lr_schedule, dtype=np.float32)
        _check(self.lib.rlx_ppo_update_f32(
            self.h, ctypes.byref(pdesc), _ptr(pparams, f), _ptr(pm, f), _ptr(pv, f), ctypes.byref(cdesc),
            _ptr(cparams, f), _ptr(cm, f), _ptr(cv, f), _ptr(states, f), _ptr(actions, f), _ptr(log_probs, f),
            _ptr(returns, f), _ptr(advantages, f), T, N, nr_epochs, minibatch_size, k, scheme, ctypes.byref(cnt),
            lr.ctypes.data_as(_F32HP), ctypes.byref(hp), _ptr(metrics_out, f), _stream()), "rlx_ppo_update_f32")
        return np.array([k[0], k[1]], dtype=np.uint32), cnt.value
