"""ctypes binding of liblmgpu.so (C ABI declared in include/lmgpu.h).

The library is built in-tree by `make -C gtsam_personal_amd/csrc` (see __graft_entry__.build()).
There is no Python/CPU fallback: if the shared object is missing, importing the hot path fails loudly.
"""
from __future__ import annotations

import ctypes as ct
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblmgpu.so")

LMGPU_OK, LMGPU_INDETERMINATE, LMGPU_INVALID, LMGPU_HIP_ERROR = 0, 1, 2, 3


class lmgpu_config(ct.Structure):
    _fields_ = [("device", ct.c_int32), ("rank", ct.c_int32), ("world_size", ct.c_int32), ("flags", ct.c_int32)]


class lmgpu_lm_params(ct.Structure):
    _fields_ = [
        ("maxIterations", ct.c_int32),
        ("relativeErrorTol", ct.c_double), ("absoluteErrorTol", ct.c_double), ("errorTol", ct.c_double),
        ("lambdaInitial", ct.c_double), ("lambdaFactor", ct.c_double), ("lambdaUpperBound", ct.c_double), ("lambdaLowerBound", ct.c_double),
        ("minModelFidelity", ct.c_double),
        ("diagonalDamping", ct.c_int32), ("useFixedLambdaFactor", ct.c_int32),
        ("minDiagonal", ct.c_double), ("maxDiagonal", ct.c_double),
    ]


class lmgpu_lm_state(ct.Structure):
    _fields_ = [("error", ct.c_double), ("lambda_", ct.c_double), ("currentFactor", ct.c_double), ("iterations", ct.c_int32),
                ("totalNumberInnerIterations", ct.c_int32)]


class lmgpu_isam2_params(ct.Structure):
    _fields_ = [("relinearizeThreshold", ct.c_double), ("relinearizeSkip", ct.c_int32), ("enableRelinearization", ct.c_int32),
                ("wildfireThreshold", ct.c_double)]


class lmgpu_isam2_result(ct.Structure):
    _fields_ = [("variablesRelinearized", ct.c_int32), ("variablesReeliminated", ct.c_int32), ("factorsRecalculated", ct.c_int32),
                ("cliques", ct.c_int32), ("batch", ct.c_int32)]


class lmgpu_isam2_update_params(ct.Structure):
    _fields_ = [("n_remove", ct.c_int32), ("removeFactorIndices", ct.POINTER(ct.c_uint64)), ("has_constrained", ct.c_int32),
                ("n_constrained", ct.c_int32), ("constrainedKeys", ct.POINTER(ct.c_uint64)), ("constrainedGroups", ct.POINTER(ct.c_int32)),
                ("n_no_relin", ct.c_int32), ("noRelinKeys", ct.POINTER(ct.c_uint64)), ("n_extra_reelim", ct.c_int32),
                ("extraReelimKeys", ct.POINTER(ct.c_uint64)), ("force_relinearize", ct.c_int32), ("forceFullSolve", ct.c_int32)]


# lmgpu_ccolamd_fn: int fn(user, n_rows, n_cols, col_ptr, row_idx, cmember, perm_out)
CCOLAMD_FN = ct.CFUNCTYPE(ct.c_int, ct.c_void_p, ct.c_int32, ct.c_int32, ct.POINTER(ct.c_int32), ct.POINTER(ct.c_int32), ct.POINTER(ct.c_int32),
                          ct.POINTER(ct.c_int32))


class lmgpu_timings(ct.Structure):
    _fields_ = [("linearize_ms", ct.c_double), ("eliminate_ms", ct.c_double), ("backsub_ms", ct.c_double), ("linear_error_ms", ct.c_double),
                ("retract_error_ms", ct.c_double), ("total_ms", ct.c_double), ("inner_iterations", ct.c_int32)]


# every symbol include/lmgpu.h declares: name -> (restype, argtypes)
_H = ct.c_void_p
_D = ct.POINTER(ct.c_double)
_I = ct.POINTER(ct.c_int32)
SYMBOLS = {
    "lmgpu_create": (ct.c_int, [ct.POINTER(lmgpu_config), ct.POINTER(_H)]),
    "lmgpu_destroy": (ct.c_int, [_H]),
    "lmgpu_last_error": (ct.c_char_p, [_H]),
    "lmgpu_last_failed_slot": (ct.c_int, [_H]),
    "lmgpu_set_variables": (ct.c_int, [_H, ct.c_int32, ct.POINTER(ct.c_uint64), _I]),
    "lmgpu_add_factor_bucket": (ct.c_int, [_H, ct.c_int32, ct.c_int32, _I, _I, _D, ct.c_int32, _D]),
    "lmgpu_add_factor_bucket_robust": (ct.c_int, [_H, ct.c_int32, ct.c_int32, _I, _I, _D, ct.c_int32, _D, ct.c_int32, ct.c_double]),
    "lmgpu_finalize_structure": (ct.c_int, [_H]),
    "lmgpu_set_values": (ct.c_int, [_H, _D]),
    "lmgpu_get_values": (ct.c_int, [_H, _D]),
    "lmgpu_save_values": (ct.c_int, [_H]),
    "lmgpu_restore_values": (ct.c_int, [_H]),
    "lmgpu_total_dim": (ct.c_int, [_H]),
    "lmgpu_total_store": (ct.c_int, [_H]),
    "lmgpu_error": (ct.c_int, [_H, _D]),
    "lmgpu_linearize": (ct.c_int, [_H]),
    "lmgpu_solve": (ct.c_int, [_H, ct.c_double, ct.c_int32, ct.c_double, ct.c_double, _D, _D, _D]),
    "lmgpu_retract": (ct.c_int, [_H, _D]),
    "lmgpu_hessian_diagonal": (ct.c_int, [_H, _D]),
    "lmgpu_lm_init": (ct.c_int, [_H, ct.POINTER(lmgpu_lm_params), ct.POINTER(lmgpu_lm_state)]),
    "lmgpu_iterate": (ct.c_int, [_H, ct.POINTER(lmgpu_lm_params), ct.POINTER(lmgpu_lm_state)]),
    "lmgpu_optimize": (ct.c_int, [_H, ct.POINTER(lmgpu_lm_params), ct.POINTER(lmgpu_lm_state)]),
    "lmgpu_gn_iterate": (ct.c_int, [_H, ct.POINTER(lmgpu_lm_state)]),
    "lmgpu_gn_optimize": (ct.c_int, [_H, ct.POINTER(lmgpu_lm_params), ct.POINTER(lmgpu_lm_state)]),
    "lmgpu_dl_iterate": (ct.c_int, [_H, ct.POINTER(lmgpu_lm_state)]),
    "lmgpu_dl_optimize": (ct.c_int, [_H, ct.POINTER(lmgpu_lm_params), ct.POINTER(lmgpu_lm_state)]),
    "lmgpu_get_timings": (ct.c_int, [_H, ct.POINTER(lmgpu_timings)]),
    "lmgpu_set_kernel_timing": (ct.c_int, [_H, ct.c_int32]),
    "lmgpu_get_kernel_times": (ct.c_int, [_H, _D, _D, ct.POINTER(ct.c_int64)]),
    "lmgpu_get_jacobian": (ct.c_int, [_H, ct.c_int32, _D, _I, _I]),
    "lmgpu_get_jacobians": (ct.c_int, [_H, _I, _I, _I, _I, ct.POINTER(ct.c_int64), _D]),
    "lmgpu_num_fronts": (ct.c_int, [_H]),
    "lmgpu_front_info": (ct.c_int, [_H, ct.c_int32, _I]),
    "lmgpu_get_front": (ct.c_int, [_H, ct.c_int32, _I, _D]),
    "lmgpu_comm_unique_id": (ct.c_int, [ct.c_char_p]),
    "lmgpu_comm_init": (ct.c_int, [_H, ct.c_char_p]),
    "lmgpu_marginal_covariance": (ct.c_int, [_H, ct.c_int32, _D]),
    "lmgpu_joint_marginal_covariance": (ct.c_int, [_H, ct.c_int32, _I, _D]),
    "lmgpu_selftest_chain_schedule": (ct.c_int, [ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.c_int]),
    "lmgpu_isam2_create": (ct.c_int, [ct.POINTER(lmgpu_config), ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.POINTER(_H)]),
    "lmgpu_isam2_destroy": (ct.c_int, [_H]),
    "lmgpu_isam2_last_error": (ct.c_char_p, [_H]),
    "lmgpu_isam2_last_failed_key": (ct.c_uint64, [_H]),
    "lmgpu_isam2_add_variables": (ct.c_int, [_H, ct.c_int32, ct.POINTER(ct.c_uint64), _I, _D]),
    "lmgpu_isam2_add_factors": (ct.c_int, [_H, ct.c_int32, ct.c_int32, ct.POINTER(ct.c_uint64), _D, ct.c_int32, _D]),
    "lmgpu_isam2_add_factors_robust": (ct.c_int, [_H, ct.c_int32, ct.c_int32, ct.POINTER(ct.c_uint64), _D, ct.c_int32, _D, ct.c_int32, ct.c_double]),
    "lmgpu_isam2_update": (ct.c_int, [_H, ct.c_int32, ct.c_void_p]),
    "lmgpu_isam2_update_with": (ct.c_int, [_H, ct.c_void_p, ct.c_void_p]),
    "lmgpu_isam2_set_relinearize_thresholds": (ct.c_int, [_H, ct.c_int32, ct.c_char_p, _I, _D]),
    "lmgpu_isam2_set_partial_relinearization_check": (ct.c_int, [_H, ct.c_int32]),
    "lmgpu_isam2_set_dogleg": (ct.c_int, [_H, ct.c_double, ct.c_double, ct.c_int32]),
    "lmgpu_isam2_get_dogleg_delta": (ct.c_double, [_H]),
    "lmgpu_isam2_set_evaluate_nonlinear_error": (ct.c_int, [_H, ct.c_int32]),
    "lmgpu_isam2_get_errors": (ct.c_int, [_H, _D, _D]),
    "lmgpu_isam2_error": (ct.c_int, [_H, ct.c_int32, _D]),
    "lmgpu_isam2_get_unused_keys": (ct.c_int, [_H, ct.POINTER(ct.c_uint64)]),
    "lmgpu_isam2_factor_exists": (ct.c_int, [_H, ct.c_int32]),
    "lmgpu_isam2_set_find_unused_factor_slots": (ct.c_int, [_H, ct.c_int32]),
    "lmgpu_isam2_marginalize_leaves": (ct.c_int, [_H, ct.c_int32, ct.POINTER(ct.c_uint64), _I, _I]),
    "lmgpu_isam2_get_marginalize_result": (ct.c_int, [_H, ct.POINTER(ct.c_uint64), ct.POINTER(ct.c_uint64)]),
    "lmgpu_isam2_get_fixed_variables": (ct.c_int, [_H, ct.POINTER(ct.c_uint64)]),
    "lmgpu_isam2_get_marginal_factor": (ct.c_int, [_H, ct.c_int32, ct.POINTER(ct.c_uint64), _I, _D]),
    "lmgpu_isam2_num_variables": (ct.c_int, [_H]),
    "lmgpu_isam2_num_factors": (ct.c_int, [_H]),
    "lmgpu_isam2_get_values": (ct.c_int, [_H, ct.c_int32, ct.POINTER(ct.c_uint64), _I, _D]),
    "lmgpu_isam2_get_value": (ct.c_int, [_H, ct.c_int32, ct.c_uint64, _I, _D]),
    "lmgpu_isam2_get_delta": (ct.c_int, [_H, _D]),
    "lmgpu_isam2_marginal_covariance": (ct.c_int, [_H, ct.c_uint64, _D]),
    "lmgpu_isam2_num_cliques": (ct.c_int, [_H]),
    "lmgpu_isam2_clique_info": (ct.c_int, [_H, ct.c_int32, _I]),
    "lmgpu_isam2_get_clique": (ct.c_int, [_H, ct.c_int32, ct.POINTER(ct.c_uint64), _D]),
    "lmgpu_peak_mfma_f64": (ct.c_int, [ct.c_int32, ct.c_int32, _D]),
    "lmgpu_peak_mfma_f64_clock": (ct.c_int, [ct.c_int32, ct.c_int32, ct.c_int32, _D, _D, _D]),
    "lmgpu_peak_hbm_copy": (ct.c_int, [ct.c_int32, ct.c_int64, ct.c_int32, _D]),
}

# test hooks: liblmgpu_test.so only (include/lmgpu.h, LMGPU_TEST_HOOKS)
TEST_SYMBOLS = {
    "lmgpu_local_group_create": (ct.c_int, [ct.c_int32, ct.POINTER(ct.c_void_p)]),
    "lmgpu_local_group_destroy": (ct.c_int, [ct.c_void_p]),
    "lmgpu_comm_init_local": (ct.c_int, [_H, ct.c_void_p]),
}

KT_NAMES = ("linearize", "lds_front", "hbm_assemble", "panel", "syrk", "backsub_hbm", "backsub_lds", "linear_error", "retract_error", "allreduce", "chain")

_lib = None
_lib_test = None
TEST_LIB_PATH = os.path.join(_HERE, "liblmgpu_test.so")


_prefer_test_library = False


def use_test_library(on: bool):
    """tests only: make load() hand out liblmgpu_test.so -- the build that reads the LMGPU_* development switches from the environment
    (A/B forms of the same arithmetic, tests/test_gpu_lookahead.py) and exports the in-process communicator.  The product library reads
    none of them."""
    global _prefer_test_library
    _prefer_test_library = bool(on)


def load(test_hooks=False):
    """Load liblmgpu.so once; raises if it has not been built (no fallback).  test_hooks=True: liblmgpu_test.so, the same library
    built with the in-process communicator the sharded-loop test needs and the development switches (never used by the product path)."""
    global _lib, _lib_test
    if test_hooks or _prefer_test_library:
        if _lib_test is None:
            if not os.path.exists(TEST_LIB_PATH):
                raise ImportError(f"{TEST_LIB_PATH} not built: run `make -C gtsam_personal_amd/csrc`")
            lib = ct.CDLL(TEST_LIB_PATH)
            for name, (res, args) in list(SYMBOLS.items()) + list(TEST_SYMBOLS.items()):
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _lib_test = lib
        return _lib_test
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(or `make -C gtsam_personal_amd/csrc`). The LM hot path has no CPU fallback.")
        lib = ct.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class LmgpuError(RuntimeError):
    pass


class IndeterminantLinearSystemException(LmgpuError):
    """mirror of gtsam/linear/linearExceptions.h; `slot` is the first frontal variable of the failing front"""

    def __init__(self, slot):
        super().__init__(f"indeterminant linear system near variable slot {slot}")
        self.slot = slot
