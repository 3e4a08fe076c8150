"""ctypes binding of libgpf_hip.so (include/gpf.h).  No CPU fallback: if the HIP library is
missing this module raises at import of the symbols, and gpf_create fails without a GPU."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GPF_LIB_OVERRIDE") or os.path.join(_HERE, "libgpf_hip.so")   # override: kernel-ablation builds only
ABI_VERSION = 1

# gpf_status
OK, ERR_INVALID_ARGUMENT, ERR_INVALID_WEIGHTS, ERR_UNKNOWN_METHOD, ERR_HIP, ERR_NO_DEVICE, ERR_STATE = range(7)
# kernel ids (gpf_kernel_id)
K_STEP, K_MAX, K_SCAN, K_SEARCH, K_GATHER, K_MOVE = range(6)
KERNEL_NAMES = {K_STEP: "k_step", K_MAX: "k_max_partial", K_SCAN: "k_scan", K_SEARCH: "k_search",
                K_GATHER: "k_gather", K_MOVE: "k_move"}


class GpfConfig(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("model", C.c_int32), ("n_params", C.c_int32),
                ("keep_prev", C.c_int32), ("params", C.POINTER(C.c_double)),
                ("n_particles", C.c_int64), ("n_global", C.c_int64), ("gid0", C.c_int64),
                ("seed", C.c_uint64), ("device", C.c_int32), ("reserved", C.c_int32),
                ("stream", C.c_void_p)]


# every symbol include/gpf.h declares: (name, restype, argtypes)
_H = C.c_void_p
_pd, _pi64, _pi32, _pu64 = C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_uint64)
SYMBOLS = [
    ("gpf_abi_version", C.c_int, []),
    ("gpf_create", C.c_int, [C.POINTER(GpfConfig), C.POINTER(_H)]),
    ("gpf_destroy", C.c_int, [_H]),
    ("gpf_last_error", C.c_char_p, [_H]),
    ("gpf_synchronize", C.c_int, [_H]),
    ("gpf_initialize", C.c_int, [_H, _pd, C.c_int32]),
    ("gpf_update", C.c_int, [_H, C.c_void_p, C.c_int32]),          # (const double*: api.py passes the address as an integer)
    ("gpf_initialize_proposal", C.c_int, [_H, _pd, C.c_int32, C.c_int32]),
    ("gpf_update_proposal", C.c_int, [_H, _pd, C.c_int32, C.c_int32]),
    ("gpf_step_ess", C.c_int, [_H, C.c_void_p, C.c_int32, C.c_double, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                               C.POINTER(C.c_int32), C.POINTER(C.c_int32), _pd]),
    ("gpf_initialize_strata", C.c_int, [_H, _pd, C.c_int32, _pd, C.c_int32, C.c_int32]),
    ("gpf_update_strata", C.c_int, [_H, _pd, C.c_int32, _pd, C.c_int32, C.c_int32]),
    ("gpf_initialize_strata_proposal", C.c_int, [_H, _pd, C.c_int32, _pd, C.c_int32, C.c_int32, C.c_int32]),
    ("gpf_resample", C.c_int, [_H, C.c_int32, C.c_double, C.c_int32, C.c_int32, _pi32]),
    ("gpf_resample_local", C.c_int, [_H, C.c_int32, C.c_int32, C.c_int32, _pi32]),
    ("gpf_resample_blocks", C.c_int, [_H, C.c_int32, C.c_int64, C.c_double, C.c_int32, C.c_double, C.c_int32, _pi32, C.POINTER(C.c_int64)]),
    ("gpf_block_resampled", C.c_int, [_H, _pi32]),
    ("gpf_block_stats", C.c_int, [_H, C.c_int64, _pd, _pd]),
    ("gpf_initialize_blocks", C.c_int, [_H, _pd, C.c_int32, C.c_int64]),
    ("gpf_update_blocks", C.c_int, [_H, _pd, C.c_int32, C.c_int64]),
    ("gpf_initialize_blocks_strata", C.c_int, [_H, _pd, C.c_int32, C.c_int64, _pd, C.c_int32, C.c_int32]),
    ("gpf_update_blocks_strata", C.c_int, [_H, _pd, C.c_int32, C.c_int64, _pd, C.c_int32, C.c_int32]),
    ("gpf_update_blocks_proposal", C.c_int, [_H, _pd, C.c_int32, C.c_int64, C.POINTER(C.c_int32), C.c_int32]),
    ("gpf_rejuvenate_blocks", C.c_int, [_H, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_uint64)]),
    ("gpf_resample_with_priorities", C.c_int, [_H, C.c_int32, _pd, C.c_int32, C.c_int32, _pi32]),
    ("gpf_rejuvenate", C.c_int, [_H, C.c_int32, C.c_int32, _pu64]),
    ("gpf_rejuvenate_proposal", C.c_int, [_H, C.c_int32, _pd, C.c_int32, C.c_int32]),
    ("gpf_rejuvenate_with_proposal", C.c_int, [_H, C.c_int32, C.c_int32, _pd, C.c_int32, C.c_int32, _pu64]),
    ("gpf_effective_sample_size", C.c_int, [_H, _pd]),
    ("gpf_log_ml_estimate", C.c_int, [_H, _pd]),
    ("gpf_get_log_weights", C.c_int, [_H, _pd, C.c_int64]),
    ("gpf_get_log_norm_weights", C.c_int, [_H, _pd, C.c_int64]),
    ("gpf_get_norm_weights", C.c_int, [_H, _pd, C.c_int64]),
    ("gpf_get_parents", C.c_int, [_H, _pi64, C.c_int64]),
    ("gpf_state_dim", C.c_int, [_H, _pi32, _pi32]),
    ("gpf_get_column", C.c_int, [_H, C.c_int32, _pd, C.c_int64]),
    ("gpf_get_rows", C.c_int, [_H, _pd, C.c_int64]),
    ("gpf_set_rows", C.c_int, [_H, _pd, C.c_int64]),
    ("gpf_set_log_weights", C.c_int, [_H, _pd, C.c_int64]),
    ("gpf_sample_unweighted", C.c_int, [_H, C.c_int64, _pd, _pi64]),
    ("gpf_mean", C.c_int, [_H, C.c_int32, _pd]),
    ("gpf_var", C.c_int, [_H, C.c_int32, _pd]),
    ("gpf_kernel_timing", C.c_int, [_H, C.c_int32, C.c_int32]),
    ("gpf_kernel_time", C.c_int, [_H, C.c_int32, _pd, _pi64]),
    ("gpf_debug_math", C.c_int, [_H, C.c_int32, _pd, _pd, C.c_int64, _pd, _pd]),
    ("gpf_debug_levels", C.c_int, [_H, C.c_int32, C.c_void_p, C.POINTER(C.c_int64)]),
    ("gpf_view_create", C.c_int, [_H, C.c_int64, C.c_int64, C.POINTER(_H)]),
    ("gpf_view_create_strided", C.c_int, [_H, C.c_int64, C.c_int64, C.c_int64, C.POINTER(_H)]),
    ("gpf_view_create_indexed", C.c_int, [_H, C.POINTER(C.c_int64), C.c_int64, C.POINTER(_H)]),
    ("gpf_n_particles", C.c_int, [_H, _pi64]),
    ("gpf_resize", C.c_int, [_H, C.c_int64, C.c_int32, C.c_double, C.c_int32, _pi32]),
    ("gpf_replicate", C.c_int, [_H, C.c_int32, C.c_int32]),
    ("gpf_dereplicate", C.c_int, [_H, C.c_int32, C.c_int32, C.c_int32]),
    ("gpf_history_enable", C.c_int, [_H, C.c_int32]),
    ("gpf_history_steps", C.c_int, [_H, _pi32]),
    ("gpf_history_column", C.c_int, [_H, C.c_int32, C.c_int32, _pd, C.c_int64]),
    ("gpf_history_mean", C.c_int, [_H, C.c_int32, C.c_int32, _pd]),
    ("gpf_history_var", C.c_int, [_H, C.c_int32, C.c_int32, _pd]),
    ("gpf_proportion", C.c_int, [_H, C.c_int32, C.c_int32, C.c_double, _pd]),
    # shard-level building blocks: device pointers are passed as integers (tensor.data_ptr())
    ("gpf_shard_weight_max", C.c_int, [_H, C.c_void_p]),
    ("gpf_shard_weight_scan", C.c_int, [_H, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    ("gpf_shard_flags", C.c_int, [_H, _pi32]),
    ("gpf_shard_residual_scan", C.c_int, [_H, C.c_void_p, C.c_int32, C.c_void_p]),
    ("gpf_shard_push_count", C.c_int, [_H, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int64)]),
    ("gpf_shard_counts", C.c_int, [_H, C.c_int32, C.POINTER(C.c_int64)]),
    ("gpf_shard_push", C.c_int, [_H, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.c_int64, C.c_void_p]),
    ("gpf_shard_commit", C.c_int, [_H, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32]),
    ("gpf_shard_lml_est", C.c_int, [_H, _pd]),
    ("gpf_comm_unique_id", C.c_int, [C.c_void_p]),
    ("gpf_comm_create", C.c_int, [_H, C.c_void_p, C.c_int32, C.c_int32]),
    ("gpf_comm_destroy", C.c_int, [_H]),
    ("gpf_comm_summary_mode", C.c_int, [_H, _pi32]),
    ("gpf_shard_step_ess", C.c_int, [_H, C.c_void_p, C.c_int32, C.c_double, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                     C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("gpf_comm_traffic", C.c_int, [_H, C.POINTER(C.c_int64), C.c_int32]),
    ("gpf_comm_set_plan", C.c_int, [_H, C.c_int32]),
    ("gpf_comm_plan", C.c_int, [_H, _pi32]),
    ("gpf_comm_calibrate", C.c_int, [_H, C.c_int64, C.c_int32, _pd]),
    ("gpf_comm_set_exchange", C.c_int, [_H, C.c_int32]),
    ("gpf_comm_exchange", C.c_int, [_H, _pi32]),
    ("gpf_phase_timing", C.c_int, [_H, C.c_int32]),
    ("gpf_phase_times", C.c_int, [_H, _pd, _pi64]),
    ("gpf_shard_resample", C.c_int, [_H, C.c_int32, C.c_int32, _pi32]),
    ("gpf_shard_resample_tempered", C.c_int, [_H, C.c_int32, C.c_double, C.c_int32, _pi32]),
    ("gpf_checkpoint_size", C.c_int, [_H, C.POINTER(C.c_int64)]),
    ("gpf_checkpoint_save", C.c_int, [_H, C.c_void_p, C.c_int64]),
    ("gpf_checkpoint_load", C.c_int, [_H, C.c_void_p, C.c_int64]),
    ("gpf_shard_resample_sorted", C.c_int, [_H, C.c_int32, _pi32]),
    ("gpf_shard_sorted_count", C.c_int, [_H, C.c_void_p, C.c_int32, C.c_int32]),
    ("gpf_shard_sorted_push", C.c_int, [_H, C.c_int32, C.c_int32, C.c_int64, C.c_void_p]),
    ("gpf_shard_effective_sample_size", C.c_int, [_H, _pd]),
    ("gpf_shard_log_ml_estimate", C.c_int, [_H, _pd]),
    # host-side scalar spec
    ("gpf_set_lazy_search", C.c_int, [_H, C.c_int32]),
    ("gpf_host_fix_K", C.c_int32, [C.c_int64]),
    ("gpf_host_gamma_E", C.c_int32, [C.c_int64]),
    ("gpf_host_div128", C.c_uint64, [C.c_uint64, C.c_uint64]),
    ("gpf_host_muldiv128", C.c_uint64, [C.c_uint64, C.c_uint64, C.c_uint64]),
    ("gpf_host_gamma_tile", C.c_uint64, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int64, C.c_int32]),
    ("gpf_host_log", C.c_double, [C.c_double]),
    ("gpf_host_lse", C.c_double, [C.c_double, C.c_uint64, C.c_int32, C.c_int32]),
    ("gpf_host_ess", C.c_double, [C.c_uint64, C.c_uint64, C.c_uint64]),
    ("gpf_host_math", None, [C.c_int32, _pd, _pd, C.c_int64, _pd, _pd]),
]

_lib = None


class GpfLibraryMissing(ImportError):
    pass


def load():
    """Load libgpf_hip.so (built by __graft_entry__.build()).  torch, when importable, is imported
    first so that both share ONE HIP runtime (torch bundles libamdhip64.so.7 under the same SONAME)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GpfLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`. "
            "The HIP extension is mandatory; there is no CPU fallback.")
    try:
        import torch  # noqa: F401  (runtime sharing only)
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        f = getattr(L, name)          # AttributeError if a declared symbol is not exported
        f.restype, f.argtypes = res, args
    if L.gpf_abi_version() != ABI_VERSION:
        raise GpfLibraryMissing("libgpf_hip.so ABI version mismatch; rebuild")
    _lib = L
    return L
