"""ctypes binding of the C-ABI in include/mppi_amd.h (libmppi_amd.so, built by _build.py).

There is no fallback: if the HIP library cannot be built or loaded, importing the engine's
compute entry points raises.  All functions take raw device addresses (torch ``data_ptr()``)
and the caller's HIP stream handle; tensors stay owned by PyTorch.
"""
import ctypes as C
import os

from . import _build

ABI_VERSION = 22

F32, F64 = 0, 1
NOISE_TNK4, NOISE_PHILOX, NOISE_ACTIONS, NOISE_KTN = 0, 1, 2, 3
E_UNSUPPORTED = -2
FORM_NONE, FORM_STREAMING, FORM_SINGLE_LAUNCH, FORM_ONCHIP = 0, 1, 2, 3      # mppi_last_command_form()
NEXT_DRAW_TORCH, NEXT_DRAW_PHILOX = 0, 1                                       # MppiProblem.next_kind
E_DIST = -4
E_GROUP_PEER = -5
MODEL_NONE, MODEL_PENDULUM, MODEL_INTEGRATOR, MODEL_LINEAR_GOAL, MODEL_MLP = 0, 1, 2, 3, 4
MODEL_CUSTOM_BASE = 100
MODEL_FLAG_EXACT_FP32 = 1
MODEL_FLAG_NO_WIDE = 2

_vp = C.c_void_p


class MppiProblem(C.Structure):
    """Mirror of ``struct MppiProblem`` (include/mppi_amd.h) -- field order is the ABI."""
    _fields_ = [
        ("K", C.c_int32), ("T", C.c_int32), ("nx", C.c_int32), ("nu", C.c_int32),
        ("S", C.c_int32), ("dtype", C.c_int32),
        ("k_offset", C.c_int64),
        ("model_id", C.c_int32), ("sigma_diagonal", C.c_int32), ("noise_abs_cost", C.c_int32),
        ("sample_null_action", C.c_int32), ("n_sampler_rows", C.c_int32),
        ("state_per_sample", C.c_int32), ("shift", C.c_int32), ("use_terminal", C.c_int32),
        ("noise_src", C.c_int32), ("u_per_command", C.c_int32), ("rollout_samples", C.c_int32),
        ("hidden", C.c_int32), ("num_envs", C.c_int32), ("noise_coloured", C.c_int32), ("model_flags", C.c_int32),
        ("lambda_", C.c_double), ("u_scale", C.c_double),
        ("seed", C.c_uint64), ("call", C.c_uint64), ("noise_pitch", C.c_int64),
        ("noise_rescale", C.c_double), ("smooth_weight", C.c_double),
        ("rollout_var_cost", C.c_double), ("rollout_var_discount", C.c_double),
        ("state", _vp), ("U", _vp), ("u_init", _vp), ("noise_mu", _vp), ("noise_L", _vp),
        ("sigma_inv", _vp), ("u_min", _vp), ("u_max", _vp), ("model_params", _vp), ("z", _vp),
        ("sampler_actions", _vp), ("W", _vp), ("theta", _vp), ("base_seq", _vp), ("process_noise_sd", _vp),
        ("cost_total", _vp), ("omega", _vp), ("cost_total_non_zero", _vp), ("U_out", _vp),
        ("action_out", _vp), ("perturbed_action", _vp), ("noise", _vp), ("pert_cost", _vp),
        ("states", _vp), ("record", _vp),
        ("workspace", _vp), ("workspace_elems", C.c_int64),
        ("onchip_spill", _vp), ("onchip_spill_elems", C.c_int64),
        # ABI 21: the next command's torch-stream draw inside this command's K3 launch
        ("next_z", _vp), ("next_seed", C.c_uint64), ("next_philox_offset", C.c_uint64), ("next_grid_blocks",
                C.c_int32), ("next_kind", C.c_int32), ("philox_rounds", C.c_int32),
        ("model_params_elems", C.c_int32),     # ABI 22
    ]


_PP = C.POINTER(MppiProblem)

# name -> (restype, argtypes); every symbol include/mppi_amd.h declares
SYMBOLS = {
    "mppi_abi_version": (C.c_int, []),
    "mppi_problem_size": (C.c_int64, []),
    "mppi_last_error": (C.c_char_p, []),
    "mppi_noise_rows4": (C.c_int64, [C.c_int32, C.c_int32]),
    "mppi_noise_pitch": (C.c_int64, [C.c_int32, C.c_int32]),
    "mppi_workspace_elems": (C.c_int64, [_PP]),
    "mppi_onchip_spill_elems": (C.c_int64, [_PP]),
    "mppi_model_supported": (C.c_int, [C.c_int32] * 5),
    "mppi_noise_fill_philox": (C.c_int, [_PP, _vp, _vp]),
    "mppi_noise_fill_philox_coloured": (C.c_int, [_PP, _vp, _vp]),
    "mppi_noise_fill_torch": (C.c_int, [_vp, C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_uint64, C.c_uint64,
            C.c_int32, _vp]),
    "mppi_noise_from_ktn": (C.c_int, [_PP, _vp, _vp, _vp]),
    "mppi_process_noise_export": (C.c_int, [_PP, _vp, _vp]),
    "mppi_kmppi_interp": (C.c_int, [_PP, _vp, _vp]),
    "mppi_rollout_cost": (C.c_int, [_PP, _vp]),
    "mppi_rollout_cost_kmppi": (C.c_int, [_PP, _vp]),
    "mppi_kmppi_shift": (C.c_int, [C.c_int32] * 4 + [_vp] * 7),
    "mppi_kmppi_trajectory": (C.c_int, [C.c_int32] * 4 + [_vp] * 4),
    "mppi_kmppi_after_update": (C.c_int, [C.c_int32] * 4 + [_vp] * 8),
    "mppi_upload_small": (C.c_int, [_vp, C.c_int64, _vp, _vp]),
    "mppi_smppi_shift": (C.c_int, [C.c_int32] * 3 + [_vp] * 3 + [C.c_double] + [_vp] * 4),
    "mppi_prepare": (C.c_int, [_PP, _vp]),
    "mppi_cost_block_min": (C.c_int, [_PP, _vp]),
    "mppi_weights_partial": (C.c_int, [_PP, _vp]),
    "mppi_finalize": (C.c_int, [_PP, C.c_int, _vp]),
    "mppi_command": (C.c_int, [_PP, C.c_int, _vp]),
    "mppi_stat_single_launch_commands": (C.c_int64, []),
    "mppi_stat_onchip_commands": (C.c_int64, []),
    "mppi_stat_onchip_pair_launches": (C.c_int64, []),
    "mppi_stat_mlp_split_launches": (C.c_int64, []),
    "mppi_last_command_form": (C.c_int, []),
    "mppi_last_next_draw": (C.c_int, []),
    "mppi_stat_kmppi_fused_rollouts": (C.c_int64, []),
    "mppi_command_kmppi": (C.c_int, [_PP, _PP, C.c_int, _vp]),
    "mppi_stat_kmppi_onchip_updates": (C.c_int64, []),
    "mppi_combine": (C.c_int, [_PP, _vp, C.c_int32, _vp]),
    "mppi_combine_ptrs": (C.c_int, [_PP, C.POINTER(C.c_void_p), C.c_int32, _vp]),
    "mppi_register_model": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _vp, _vp]),
    "mppi_dist_available": (C.c_int, []),
    "mppi_dist_unique_id": (C.c_int, [_vp]),
    "mppi_dist_init": (C.c_int, [_vp, C.c_int32, C.c_int32, C.POINTER(_vp)]),
    "mppi_dist_destroy": (C.c_int, [_vp]),
    "mppi_exchange_combine": (C.c_int, [_PP, _vp, _vp, C.c_int32, _vp]),
    "mppi_command_sharded": (C.c_int, [_PP, _vp, _vp, C.c_int32, _vp]),
    "mppi_dist_init_all": (C.c_int, [C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_void_p)]),
    "mppi_exchange_combine_all": (C.c_int, [C.c_int32, C.POINTER(C.c_int32), C.POINTER(_PP), C.POINTER(C.c_void_p),
            C.POINTER(C.c_void_p),
                                            C.POINTER(C.c_void_p)]),
    "mppi_group_create": (C.c_int, [C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "mppi_group_destroy": (C.c_int, [_vp]),
    "mppi_group_size": (C.c_int, [_vp]),
    "mppi_group_broadcast": (C.c_int, [_vp, _vp, C.c_int64, C.POINTER(C.c_void_p), _vp]),
    "mppi_group_submit": (C.c_int, [_vp, C.c_int32, _PP, _PP, _vp, _vp]),
    "mppi_group_wait": (C.c_int, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "mppi_group_abort": (C.c_int, [_vp]),
    "mppi_profile_enable": (C.c_int, [C.c_int]),
    "mppi_profile_read": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "mppi_profile_read2": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64),
            C.POINTER(C.c_int64)]),
    "mppi_profile_read_launches": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int64,
            C.POINTER(C.c_int64)]),
}

_lib = None


def lib():
    """Load (building first if the in-tree library is missing or stale) libmppi_amd.so."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB
    if not _build.is_current():
        # one builder at a time (torch.distributed.run starts N ranks at once): the others wait on
        # the lock and then find the library current
        import fcntl
        with open(path + ".lock", "w") as lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            try:
                if not _build.is_current():
                    _build.build(verbose=False)
            except Exception as e:  # stale-but-present library on a box without hipcc is still usable
                if not os.path.exists(path):
                    raise RuntimeError(
                        "pytorch_mppi_amd: the HIP engine library is missing and could not be built "
                        f"({e}); there is no CPU fallback") from e
            finally:
                fcntl.flock(lk, fcntl.LOCK_UN)
    try:
        l = C.CDLL(path)
    except OSError as e:
        raise RuntimeError(f"pytorch_mppi_amd: cannot load {path}: {e}; there is no CPU fallback") from e
    for name, (res, args) in SYMBOLS.items():
        f = getattr(l, name)
        f.restype = res
        f.argtypes = args
    v = l.mppi_abi_version()
    if v != ABI_VERSION:
        raise RuntimeError(f"pytorch_mppi_amd: ABI mismatch (library {v}, binding {ABI_VERSION}); rebuild")
    if l.mppi_problem_size() != C.sizeof(MppiProblem):
        raise RuntimeError(f"pytorch_mppi_amd: struct MppiProblem mismatch (library {l.mppi_problem_size()} B, "
                           f"binding {C.sizeof(MppiProblem)} B)")
    _lib = l
    return l


def check(code, what):
    if code != 0:
        msg = lib().mppi_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (code {code}): {msg}")


def profile_read_launches(capacity=8192):
    """Per-launch K1 timings since the last read: (device_span_us[], dispatch_us[]) as Python lists;
    dispatch_us[i] is None where launch i was not sampled (include/mppi_amd.h, measurement hooks)."""
    dev = (C.c_double * capacity)()
    disp = (C.c_double * capacity)()
    n = C.c_int64(0)
    check(lib().mppi_profile_read_launches(dev, disp, capacity, C.byref(n)), "mppi_profile_read_launches")
    m = min(int(n.value), capacity)
    return [dev[i] for i in range(m)], [disp[i] if disp[i] >= 0 else None for i in range(m)]


# results of the pure geometry / capability queries (a ctypes call costs ~1 us; these sit on every command's path)
_PURE = {}


def noise_rows4(T, nu):
    k = ("r4", T, nu)
    v = _PURE.get(k)
    if v is None:
        v = _PURE[k] = int(lib().mppi_noise_rows4(int(T), int(nu)))
    return v


def noise_pitch(K, dtype_code):
    """Row pitch (samples) of a TNK4 noise array for K samples (include/mppi_amd.h)."""
    return int(lib().mppi_noise_pitch(int(K), int(dtype_code)))


def model_supported(model_id, nx, nu, dtype_code, hidden=0):
    k = ("ms", model_id, nx, nu, dtype_code, hidden)
    v = _PURE.get(k)
    if v is None:
        v = bool(lib().mppi_model_supported(int(model_id), int(nx), int(nu), int(dtype_code), int(hidden)))
        if model_id < MODEL_CUSTOM_BASE:
            _PURE[k] = v            # (run-time registered models come and go: not cached)
    return v
