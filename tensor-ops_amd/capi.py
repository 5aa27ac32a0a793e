"""ctypes view of include/tensorops_hip.h.  Loading fails loudly when the HIP
library has not been built -- there is no fallback implementation."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# (TOPS_HIP_LIB: measurement tooling loads a development library built beside the product one, tools/build_ab_lib.py)
LIB_PATH = os.path.abspath(os.environ.get("TOPS_HIP_LIB") or os.path.join(HERE, "libtensorops_hip.so"))

c_tensor = C.c_void_p
c_expr = C.c_void_p
c_graph = C.c_void_p
i64p = C.POINTER(C.c_int64)
TO_F32 = 0
TO_F64 = 1

# name -> argtypes ; every function returns int32 status except to_last_error
SIGNATURES = {
    "to_init": [C.c_int],
    "to_shutdown": [],
    "to_device_count": [C.POINTER(C.c_int)],
    "to_set_stream": [C.c_void_p],
    "to_get_stream": [C.POINTER(C.c_void_p)],
    "to_sync": [],
    "to_stats": [i64p, i64p, i64p],
    "to_build_info": [C.POINTER(C.c_int)],
    "to_alloc": [C.c_int, C.c_int, i64p, C.c_int64, C.POINTER(c_tensor)],
    "to_wrap": [C.c_void_p, C.c_int, C.c_int, i64p, C.c_int64, C.POINTER(c_tensor)],
    "to_retain": [c_tensor],
    "to_release": [c_tensor],
    "to_shape": [c_tensor, C.POINTER(C.c_int), i64p, i64p],
    "to_dtype": [c_tensor, C.POINTER(C.c_int)],
    "to_set_default_dtype": [C.c_int],
    "to_default_dtype": [C.POINTER(C.c_int)],
    "to_is_contiguous": [c_tensor, C.POINTER(C.c_int)],
    "to_data_ptr": [c_tensor, C.POINTER(C.c_void_p)],
    "to_upload": [c_tensor, C.c_void_p, C.c_int64],
    "to_download": [c_tensor, C.c_void_p, C.c_int64],
    "to_from_host": [C.c_int, C.c_int, i64p, C.c_int64, C.c_void_p, C.POINTER(c_tensor)],
    "to_fill": [C.c_int, C.c_int, i64p, C.c_int64, C.c_double, C.POINTER(c_tensor)],
    "to_rand": [C.c_int, C.c_int, i64p, C.c_int64, C.c_int, C.c_double, C.c_double, C.c_uint64,
                C.POINTER(c_tensor)],
    "to_gmul": [C.c_int, C.c_int, C.c_int, c_tensor, c_tensor, C.POINTER(c_tensor)],
    "to_lift": [c_expr, C.c_int, C.POINTER(c_tensor), C.POINTER(c_tensor)],
    "to_sum": [C.c_int, C.POINTER(c_tensor), C.c_int, i64p, C.POINTER(c_tensor)],
    "to_scale": [C.c_double, c_tensor, C.POINTER(c_tensor)],
    "to_transp": [c_tensor, C.POINTER(c_tensor)],
    "to_sum_rows": [c_tensor, C.POINTER(c_tensor)],
    "to_map_rows_const": [C.c_int, c_tensor, c_tensor, C.POINTER(c_tensor)],
    "to_slice": [c_tensor, C.c_int, i64p, C.POINTER(c_tensor)],
    "to_stack": [C.c_int, i64p, C.POINTER(c_tensor), C.POINTER(c_tensor)],
    "to_diag": [C.c_int, c_tensor, C.POINTER(c_tensor)],
    "to_get_diag": [c_tensor, C.POINTER(c_tensor)],
    "to_index": [c_tensor, i64p, C.c_int64, C.POINTER(C.c_double)],
    "to_arg_max": [c_tensor, i64p],
    "to_arg_min": [c_tensor, i64p],
    "to_one_hot": [C.c_int, C.c_int64, C.c_double, C.c_double, C.c_int64, i64p, C.POINTER(c_tensor)],
    "to_blas_axpy": [C.c_double, c_tensor, c_tensor, C.POINTER(c_tensor)],
    "to_blas_dot": [c_tensor, c_tensor, C.POINTER(C.c_double)],
    "to_blas_ger": [c_tensor, c_tensor, C.POINTER(c_tensor)],
    "to_blas_gemv": [C.c_double, c_tensor, c_tensor, C.c_double, c_tensor, C.POINTER(c_tensor)],
    "to_blas_gemm": [C.c_double, c_tensor, c_tensor, C.c_double, c_tensor, C.POINTER(c_tensor)],
    "to_blas_scale": [C.c_double, c_tensor, C.POINTER(c_tensor)],
    "to_blas_add": [c_tensor, c_tensor, C.POINTER(c_tensor)],
    "to_blas_index_row": [C.c_int64, c_tensor, C.POINTER(c_tensor)],
    "to_blas_transp": [c_tensor, C.POINTER(c_tensor)],
    "to_blas_eye": [C.c_int, C.c_int64, C.POINTER(c_tensor)],
    "to_blas_trace": [c_tensor, C.POINTER(C.c_double)],
    "to_blas_diag": [c_tensor, C.POINTER(c_tensor)],
    "to_blas_get_diag": [c_tensor, C.POINTER(c_tensor)],
    "to_blas_sum": [c_tensor, C.POINTER(C.c_double)],
    "to_expr_compile": [C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_double),
                        C.POINTER(c_expr)],
    "to_expr_release": [c_expr],
    "to_expr_kind": [c_expr, C.POINTER(C.c_int)],
    "to_batch_sum": [c_tensor, C.POINTER(c_tensor)],
    "to_batch_bcast": [c_tensor, C.c_int64, C.POINTER(c_tensor)],
    "to_batch_slice": [c_tensor, C.c_int64, C.c_int64, C.POINTER(c_tensor)],
    "to_batch_gather": [c_tensor, C.c_int64, i64p, C.POINTER(c_tensor)],
    "to_batch_select": [c_tensor, C.c_int64, C.POINTER(c_tensor)],
    "to_gmul_batch_sum": [C.c_int, C.c_int, C.c_int, c_tensor, c_tensor, C.POINTER(c_tensor)],
    "to_memo_begin": [],
    "to_memo_end": [],
    "to_force": [c_tensor],
    "to_force_many": [C.c_int, C.POINTER(c_tensor)],
    "to_set_lazy": [C.c_int, C.POINTER(C.c_int)],
    "to_set_loss_head_match": [C.c_int, C.POINTER(C.c_int)],
    "to_lazy_stats": [i64p, i64p, i64p, i64p],
    "to_lazy_time": [i64p, i64p],
    "to_api_time": [i64p, i64p],
    "to_transfer_stats": [i64p, i64p, i64p, i64p],
    "to_plan_cache_stats": [i64p, i64p, i64p],
    "to_plan_cache_clear": [],
    "to_graph_begin": [],
    "to_graph_end": [C.POINTER(c_graph)],
    "to_graph_launch": [c_graph],
    "to_graph_info": [c_graph, C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "to_graph_release": [c_graph],
    "to_sgd_step_inplace": [c_tensor, c_tensor, C.c_double],
    "to_comm_unique_id": [C.c_void_p],
    "to_comm_init": [C.c_int, C.c_int, C.c_void_p],
    "to_comm_allreduce_sum": [c_tensor],
    "to_comm_world": [C.POINTER(C.c_int)],
    "to_comm_shutdown": [],
    "to_p2p_create": [C.c_int64, C.c_int, C.c_int, C.c_void_p],
    "to_p2p_connect": [C.c_int, C.c_void_p],
    "to_p2p_allreduce_sum": [c_tensor],
    "to_p2p_allreduce_sgd": [c_tensor, c_tensor, C.c_double, C.c_int],
    "to_p2p_status": [C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "to_p2p_shutdown": [],
    "to_copy_into_many": [C.c_int, C.POINTER(c_tensor), C.POINTER(c_tensor)],
    "to_copy_into": [c_tensor, c_tensor],
    "to_fflayer_stack_grad": [C.c_int, C.POINTER(c_tensor), C.POINTER(c_tensor), C.c_int, C.c_int, C.c_int,
                              c_tensor, c_tensor, C.POINTER(c_tensor), C.POINTER(c_tensor), c_tensor],
    "to_fflayer_stack_sgd": [C.c_int, C.POINTER(c_tensor), C.POINTER(c_tensor), C.c_int, C.c_int, C.c_int,
                             c_tensor, c_tensor, C.c_double, c_tensor],
    "to_fflayer_stack_online_sgd": [C.c_int, C.POINTER(c_tensor), C.POINTER(c_tensor), C.c_int, C.c_int, C.c_int,
                                    c_tensor, c_tensor, C.c_int64, i64p, C.c_double],
    "to_graph_online_sgd": [c_graph, c_tensor, c_tensor, c_tensor, c_tensor, C.c_int64, i64p, C.POINTER(C.c_int)],
    "to_online_sgd_stats": [i64p, i64p],
    "to_timer_start": [],
    "to_timer_stop": [C.POINTER(C.c_float)],
}

_lib = None


class TensorOpsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("tensorops_hip error %d: %s" % (code, msg))
        self.code = code


def lib():
    """Load libtensorops_hip.so (built by build.py).  Raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libtensorops_hip.so is not built (run `python tensor-ops_amd/build.py`); "
                "this backend has no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = C.c_int32
        L.to_last_error.argtypes = []
        L.to_last_error.restype = C.c_char_p
        _lib = L
    return _lib


def check(status):
    if status != 0:
        raise TensorOpsError(status, lib().to_last_error().decode(errors="replace"))


def dims_arr(dims):
    dims = [int(d) for d in dims]
    return (C.c_int64 * max(len(dims), 1))(*dims), len(dims)
