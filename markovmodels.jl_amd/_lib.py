"""ctypes binding of libmarkovmodels_amd.so (C ABI: include/markovmodels_amd.h).

There is NO fallback: if the HIP library has not been built the import fails
loudly (run ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C markovmodels.jl_amd/csrc``).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MM_AMD_LIB", os.path.join(_HERE, "libmarkovmodels_amd.so"))  # override: diagnostic builds

MM_OK = 0
MM_LOG, MM_TROPICAL, MM_PROB = 0, 1, 2
MM_CSC, MM_CSR = 0, 1
MM_EXACT_AUTO, MM_EXACT_F32_FIRST, MM_EXACT_F64_FIRST = 0, 1, 2
MM_MARKS_DECIDE, MM_MARKS_KEEP = 0, 1
MM_ABI_VERSION = 4  # include/markovmodels_amd.h: what the argtypes below were written against
SEMIRING_ID = {"log": MM_LOG, "tropical": MM_TROPICAL, "prob": MM_PROB}

#: every symbol include/markovmodels_amd.h declares
SYMBOLS = [
    "mm_abi_version",
    "mm_last_error",
    "mm_fsm_create",
    "mm_fsm_create_many",
    "mm_fsm_destroy",
    "mm_fsm_info",
    "mm_batch_create",
    "mm_batch_destroy",
    "mm_batch_total_states",
    "mm_batch_workspace_bytes",
    "mm_batch_kernels",
    "mm_batch_reserve",
    "mm_batch_reserve_ex",
    "mm_batch_set_deterministic",
    "mm_batch_set_posterior_floor",
    "mm_batch_last_redo_count",
    "mm_batch_last_fallback_count",
    "mm_batch_last_exact_first",
    "mm_batch_team_xcd_stats",
    "mm_batch_set_exact_policy",
    "mm_batch_set_mark_policy",
    "mm_batch_set_gamma_mode",
    "mm_spmv",
    "mm_spmm",
    "mm_svdv",
    "mm_pdfposteriors_f32",
    "mm_pdfposteriors_ex",
    "mm_statemap_create",
    "mm_statemap_destroy",
    "mm_alpharecursion_f32",
    "mm_betarecursion_f32",
    "mm_maxstateposteriors_f32",
    "mm_viterbi_f32",
    "mm_totalsum_f32",
    "mm_set_rccl",
    "mm_allreduce_logz",
    "mm_allgather_ttl",
    "mm_debug_packed_product",
    "mm_debug_quad_product",
    "mm_debug_row_product",
    "mm_debug_row_product_ex",
    "mm_debug_split_product",
    "mm_debug_wave_product",
    "mm_debug_stream_product",
    "mm_debug_stream_team_product",
    "mm_debug_reach_distance",
]


class MarkovModelsAMDError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"markovmodels_amd error {code}: {msg}")
        self.code = code


class DimensionMismatch(MarkovModelsAMDError):
    """The reference's DimensionMismatch (src/linalg.jl:166-167)."""


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP engine is not built and there is no CPU fallback. "
            "Build it with `make -C markovmodels.jl_amd/csrc` (hipcc --offload-arch=gfx950)."
        )
    # One HIP runtime per process: when torch is present its bundled libamdhip64 must be
    # the instance the engine binds to (torch tensors and streams are handed to the engine).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    i64, i32, vp, fp = C.c_int64, C.c_int32, C.c_void_p, C.c_void_p
    lib.mm_abi_version.restype = C.c_int
    if lib.mm_abi_version() != MM_ABI_VERSION:  # (a stale build: every call below would pass the wrong argument lists)
        raise ImportError(f"{LIB_PATH} has ABI version {lib.mm_abi_version()}, this binding was written against {MM_ABI_VERSION}: rebuild it")
    lib.mm_last_error.restype = C.c_char_p
    lib.mm_fsm_create.restype = C.c_int
    lib.mm_fsm_create.argtypes = [C.c_int, i64, i64, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, i64, vp, vp, vp,
                                  i32, C.POINTER(vp)]
    lib.mm_fsm_create_many.restype = C.c_int
    lib.mm_fsm_create_many.argtypes = [i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, vp]
    lib.mm_fsm_destroy.restype = C.c_int
    lib.mm_fsm_destroy.argtypes = [vp]
    lib.mm_fsm_info.restype = C.c_int
    lib.mm_fsm_info.argtypes = [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i32), C.POINTER(i64), C.POINTER(i64)]
    lib.mm_batch_create.restype = C.c_int
    lib.mm_batch_create.argtypes = [C.POINTER(vp), i64, C.POINTER(vp)]
    lib.mm_batch_destroy.restype = C.c_int
    lib.mm_batch_destroy.argtypes = [vp]
    lib.mm_batch_total_states.restype = i64
    lib.mm_batch_total_states.argtypes = [vp]
    lib.mm_batch_workspace_bytes.restype = C.c_size_t
    lib.mm_batch_workspace_bytes.argtypes = [vp, i64]
    lib.mm_batch_reserve.restype = C.c_int
    lib.mm_batch_reserve.argtypes = [vp, i64]
    lib.mm_batch_set_deterministic.restype = C.c_int
    lib.mm_batch_set_deterministic.argtypes = [vp, C.c_int]
    lib.mm_batch_set_posterior_floor.restype = C.c_int
    lib.mm_batch_set_posterior_floor.argtypes = [vp, C.c_float]
    lib.mm_batch_last_redo_count.restype = C.c_int
    lib.mm_batch_last_redo_count.argtypes = [vp, vp, C.POINTER(i64)]
    lib.mm_batch_last_fallback_count.restype = C.c_int
    lib.mm_batch_last_fallback_count.argtypes = [vp, vp, C.POINTER(i64)]
    lib.mm_batch_last_exact_first.restype = C.c_int
    lib.mm_batch_last_exact_first.argtypes = [vp]
    lib.mm_batch_team_xcd_stats.restype = C.c_int
    lib.mm_batch_team_xcd_stats.argtypes = [vp, vp]
    lib.mm_batch_set_exact_policy.restype = C.c_int
    lib.mm_batch_set_exact_policy.argtypes = [vp, C.c_int]
    lib.mm_batch_set_mark_policy.restype = C.c_int
    lib.mm_batch_set_mark_policy.argtypes = [vp, C.c_int]
    lib.mm_batch_set_gamma_mode.restype = C.c_int
    lib.mm_batch_set_gamma_mode.argtypes = [vp, C.c_int, C.c_float]
    lib.mm_spmv.restype = C.c_int
    lib.mm_spmv.argtypes = [C.c_int, C.c_int, i64, i64, i64, vp, vp, C.c_int, vp, vp, i64, vp, i64, vp]
    lib.mm_spmm.restype = C.c_int
    lib.mm_spmm.argtypes = [C.c_int, C.c_int, i64, i64, i64, vp, vp, C.c_int, vp, vp, i64, i64, i64, vp, i64, i64, i64, C.c_double, vp]
    lib.mm_svdv.restype = C.c_int
    lib.mm_svdv.argtypes = [C.c_int, C.c_int, C.c_int, i64, i64, vp, C.c_int, vp, vp, i64, vp, i64, vp]
    lib.mm_batch_kernels.restype = C.c_int
    lib.mm_batch_kernels.argtypes = [vp, C.c_int, C.c_char_p, C.c_size_t]
    lib.mm_statemap_create.restype = C.c_int
    lib.mm_statemap_create.argtypes = [C.c_int, i64, i32, i64, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.POINTER(vp)]
    lib.mm_statemap_destroy.restype = C.c_int
    lib.mm_statemap_destroy.argtypes = [vp]
    lib.mm_pdfposteriors_ex.restype = C.c_int
    lib.mm_pdfposteriors_ex.argtypes = [vp, vp, C.c_int, i32, vp, i64, i64, i64, vp, i64, i64, i64, vp, vp]
    lib.mm_batch_reserve_ex.restype = C.c_int
    lib.mm_batch_reserve_ex.argtypes = [vp, C.c_int, i64]
    lib.mm_pdfposteriors_f32.restype = C.c_int
    lib.mm_pdfposteriors_f32.argtypes = [vp, fp, i64, i64, vp, i64, fp, i64, i64, i64, fp, vp]
    for name in ("mm_alpharecursion_f32", "mm_betarecursion_f32", "mm_maxstateposteriors_f32"):
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = [vp, fp, i64, i64, vp, i64, fp, i64, vp]
    lib.mm_viterbi_f32.restype = C.c_int
    lib.mm_viterbi_f32.argtypes = [vp, fp, i64, i64, vp, i64, vp, i64, fp, vp, i64, vp]
    lib.mm_totalsum_f32.restype = C.c_int
    lib.mm_totalsum_f32.argtypes = [vp, i64, C.c_int, fp, vp]
    lib.mm_set_rccl.restype = C.c_int
    lib.mm_set_rccl.argtypes = [vp]
    lib.mm_allreduce_logz.restype = C.c_int
    lib.mm_allreduce_logz.argtypes = [vp, vp, i64, vp, vp]
    lib.mm_allgather_ttl.restype = C.c_int
    lib.mm_allgather_ttl.argtypes = [vp, vp, i64, vp, vp]
    lib.mm_debug_packed_product.restype = C.c_int
    lib.mm_debug_packed_product.argtypes = [vp, C.c_int, vp, vp, vp]
    lib.mm_debug_reach_distance.restype = C.c_int
    lib.mm_debug_reach_distance.argtypes = [vp, C.c_int, vp]
    lib.mm_debug_quad_product.restype = C.c_int
    lib.mm_debug_quad_product.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp]
    lib.mm_debug_row_product.restype = C.c_int
    lib.mm_debug_row_product.argtypes = [vp, C.c_int, vp, vp, vp]
    lib.mm_debug_wave_product.restype = C.c_int
    lib.mm_debug_wave_product.argtypes = [vp, C.c_int, vp, vp, vp]
    lib.mm_debug_stream_product.restype = C.c_int
    lib.mm_debug_stream_product.argtypes = [vp, C.c_int, vp, vp, vp]
    lib.mm_debug_stream_team_product.restype = C.c_int
    lib.mm_debug_stream_team_product.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp]
    lib.mm_debug_split_product.restype = C.c_int
    lib.mm_debug_split_product.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp]
    lib.mm_debug_row_product_ex.restype = C.c_int
    lib.mm_debug_row_product_ex.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp]
    return lib


lib = _load()


def check(rc: int):
    if rc != MM_OK:
        msg = lib.mm_last_error().decode("utf-8", "replace")
        if rc == -2:
            raise DimensionMismatch(rc, msg)
        raise MarkovModelsAMDError(rc, msg)
