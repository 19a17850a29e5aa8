"""ctypes binding of libmtc.so (include/mtc.h).  No fallback: if the HIP library is missing or a
call fails, this raises -- the product path never computes on the CPU."""
from __future__ import annotations

import ctypes as C
import os

MTC_F32, MTC_BF16 = 0, 1
MTC_MAX_PREFIX = 16
_LIB = None
_LIB_PATH = os.environ.get("MTC_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmtc.so")


class MtcError(RuntimeError):
    pass


class Desc(C.Structure):
    _fields_ = [("B", C.c_int32), ("A_s", C.c_int32), ("A_t", C.c_int32), ("D", C.c_int32),
                ("G", C.c_int32), ("prefix", C.c_int32 * MTC_MAX_PREFIX), ("dtype", C.c_int32),
                ("unit_norm", C.c_int32)]


class AdamHyper(C.Structure):
    _fields_ = [("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("step", C.c_int32), ("pad", C.c_int32)]


SELECT_STATE_BYTES = 64
P = C.c_void_p
DP = C.POINTER(Desc)
HP = C.POINTER(AdamHyper)

# name -> (restype, argtypes); must list every symbol include/mtc.h declares
PROTOTYPES = {
    "mtc_last_error": (C.c_char_p, []),
    "mtc_version": (C.c_int, []),
    "mtc_row_normalize": (C.c_int, [DP, P, P, P, P, P]),
    "mtc_encode_fwd": (C.c_int, [DP, P, P, P, P, P]),
    "mtc_encode_marks_bytes": (C.c_size_t, [DP]),
    "mtc_encode_fwd_marked": (C.c_int, [DP, P, P, P, P, P, P, P]),
    "mtc_encode_threshold_infer": (C.c_int, [DP, P, P, P, P, P, P]),
    "mtc_encode_set_tile": (C.c_int, [C.c_int]),
    "mtc_select_levels": (C.c_int, [DP]),
    "mtc_select_init": (C.c_int, [P, C.c_uint32, P]),
    "mtc_select_hist": (C.c_int, [DP, P, C.c_int, P, P, P]),
    "mtc_select_scan": (C.c_int, [DP, C.c_int, P, P, P]),
    "mtc_select_rowsub": (C.c_int, [DP, P, P, P, P, P]),
    "mtc_select_finish_rows": (C.c_int, [DP, P, P, P, P, P, P, P]),
    "mtc_select_count": (C.c_int, [DP, P, P, P, P, P, P]),
    "mtc_select_offsets": (C.c_int, [DP, P, P, P, P, P, P, P]),
    "mtc_select_fill": (C.c_int, [DP, P, P, P, P, P, P, P, P]),
    "mtc_select_workspace_bytes": (C.c_size_t, [DP]),
    "mtc_select_global_topk": (C.c_int, [DP, P, C.c_uint32, P, P, C.c_size_t, P, P, P, P, P]),
    "mtc_select_carry_workspace_bytes": (C.c_size_t, [DP]),
    "mtc_select_global_topk_carry": (C.c_int, [DP, P, C.c_uint32, P, C.c_size_t, P, P, P, P, P, P]),
    "mtc_select_global_topk_carry_ema": (C.c_int, [DP, P, C.c_uint32, P, C.c_size_t, P, P, P, P, P, P, C.c_float, P]),
    "mtc_select_global_topk_carry_marked": (C.c_int, [DP, P, C.c_uint32, P, C.c_size_t, P, P, P, P, P, P, C.c_float, P, P]),
    "mtc_select_offsets_sharded": (C.c_int, [DP, P, P, C.c_int32, C.c_int32, C.c_int32, P, P, P, P, P]),
    "mtc_select_pack_minmax": (C.c_int, [P, P, P]),
    "mtc_select_sharded_workspace_bytes": (C.c_size_t, [DP]),
    "mtc_select_sharded_xbuf_words": (C.c_size_t, []),
    "mtc_select_sharded_full_bins": (C.c_size_t, []),
    "mtc_select_sharded_begin": (C.c_int, [DP, P, C.c_uint32, P, C.c_size_t, P, P, P, P, P, P, P]),
    "mtc_select_sharded_window": (C.c_int, [DP, P, C.c_uint32, P, P, P, P, P]),
    "mtc_select_sharded_counts": (C.c_int, [DP, P, C.c_uint32, P, P, P, P, P, P]),
    "mtc_select_sharded_finish": (C.c_int, [DP, P, P, P, P, P, C.c_int32, C.c_int32, P, P, P, P, C.c_float, P]),
    "mtc_threshold_ema_global": (C.c_int, [DP, P, P, C.c_float, P]),
    "mtc_shard_nnz_abs": (C.c_int, [DP, P, P, P, P]),
    "mtc_threshold_ema": (C.c_int, [DP, P, P, C.c_float, P]),
    "mtc_decode_fwd": (C.c_int, [DP, P, P, P, P, P, P, P, P, P, P, P, P, P, C.c_float, P]),
    "mtc_decode_partial": (C.c_int, [DP, P, P, P, P, C.c_int32, P, P, P, P]),
    "mtc_decode_partial_strided": (C.c_int, [DP, P, P, P, P, C.c_int32, C.c_int64, P, P, P, P]),
    "mtc_decode_finish": (C.c_int, [DP, C.c_int32, P, P, P, P, P, P, P, P, P, P, P]),
    "mtc_reduce_scalars": (C.c_int, [DP, P, P, P, P, C.c_int32, C.c_float, C.c_float, P, P]),
    "mtc_decode_bwd_vals": (C.c_int, [DP, P, P, P, P, C.c_float, P, P]),
    "mtc_csc_workspace_bytes": (C.c_size_t, [DP]),
    "mtc_build_csc": (C.c_int, [DP, P, P, P, C.c_size_t, P, P, P, P, P]),
    "mtc_build_csc_part": (C.c_int, [DP, P, P, P, C.c_size_t, P, P, P, P, C.c_int32, P]),
    "mtc_backward_sparse": (C.c_int, [DP, P, P, P, P, P, P, P, P, P, P, P, C.c_int, P]),
    "mtc_colsum_workspace_bytes": (C.c_size_t, [DP]),
    "mtc_colsum": (C.c_int, [DP, P, P, P, C.c_size_t, P, C.c_int, P]),
    "mtc_colsum_static": (C.c_int, [DP, P, P, P, C.c_int, P]),
    "mtc_unitnorm_adam_rows": (C.c_int, [DP, P, P, P, P, HP, C.c_int, P]),
    "mtc_adam_flat": (C.c_int, [C.c_int32, C.c_size_t, P, P, P, P, HP, P]),
    "mtc_fused_backward_adam_dec": (C.c_int, [DP, P, P, P, P, P, P, P, P, P, HP, P, C.c_float, P]),
    "mtc_fused_backward_adam_enc": (C.c_int, [DP, P, P, P, P, P, P, P, P, P, P, P, P, HP, P]),
    "mtc_fused_backward_adam_dec_extra": (C.c_int, [DP, P, P, P, P, P, P, P, P, P, HP, P, C.c_float, P, P, C.c_int32, P]),
    "mtc_fused_backward_adam_enc_extra": (C.c_int, [DP, P, P, P, P, P, P, P, P, P, P, P, P, HP, P, P, P, C.c_int32, P]),
    "mtc_fused_set_sliced": (C.c_int, [C.c_int]),
    "mtc_diag_sliced_decoder_bound": (C.c_int, [DP, P, P, P, P, P, P, P, P, C.c_int32, P]),
    "mtc_comm_unique_id": (C.c_int, [P]),
    "mtc_comm_init": (C.c_int, [C.POINTER(P), C.c_int32, C.c_int32, P]),
    "mtc_comm_destroy": (C.c_int, [P]),
    "mtc_comm_world": (C.c_int, [P]),
    "mtc_comm_rank": (C.c_int, [P]),
    "mtc_comm_all_reduce": (C.c_int, [P, P, P, C.c_size_t, C.c_int32, C.c_int32, P]),
    "mtc_comm_reduce_scatter": (C.c_int, [P, P, P, C.c_size_t, C.c_int32, C.c_int32, P]),
    "mtc_comm_all_gather": (C.c_int, [P, P, P, C.c_size_t, C.c_int32, P]),
    "mtc_unitnorm_project": (C.c_int, [DP, P, P, P]),
    "mtc_aux_dead_list": (C.c_int, [DP, P, C.c_float, P, P, P, P, P]),
    "mtc_gemm_nt": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, P, C.c_int64, P, P, C.c_int64, P, C.c_int64, C.c_int32, P]),
    "mtc_transpose_rows": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, P, C.c_int64, P, P, C.c_int64, P]),
    "mtc_aux_mask_rowsum": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, P, C.c_int64, P, C.c_int64, P, P]),
    "mtc_rows_scatter_add": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, P, P, C.c_int64, P, C.c_int64, C.c_int32, P, P]),
    "mtc_aux_select": (C.c_int, [DP, P, P, P, C.c_int32, P, P, P, P]),
    "mtc_aux_fill": (C.c_int, [DP, P, P, P, P, P, P, P, P, P]),
    "mtc_aux_fwd": (C.c_int, [DP, P, P, P, P, P, P, P, C.c_float, P, P, P, P, P]),
    "mtc_aux_dense_fill": (C.c_int, [DP, P, P, P, P, P, C.c_int32, P, P]),
    "mtc_aux_select_dense": (C.c_int, [DP, P, P, P, C.c_int32, C.c_int32, P, P, P, P, P]),
    "mtc_aux_residual": (C.c_int, [DP, P, P, P, P, C.c_float, P, P, P, P, P]),
    "mtc_aux_mask": (C.c_int, [C.c_int32, C.c_uint64, P, P, P]),
    "mtc_aux_row_candidates": (C.c_int, [DP, P, P, P, P, P, C.c_int32, P, P]),
    "mtc_aux_count_at": (C.c_int, [DP, P, P, P, P, P, P, P]),
    "mtc_aux_count_at_groups": (C.c_int, [DP, P, P, P, P, P, P, P]),
    "mtc_densify": (C.c_int, [DP, P, P, P, C.c_int32, P, P]),
    "mtc_copy16": (C.c_int, [P, P, C.c_size_t, P]),
    "mtc_gather_row_pairs": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, P, P, P, P, P, P, P]),
    "mtc_collect_top_samples": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, P, P, P, P, P, P, C.c_float,
                                          C.c_int32, C.c_int32, P, P, P, P, P, P, P]),
}


def lib_path() -> str:
    return _LIB_PATH


def load() -> C.CDLL:
    """Load libmtc.so (built in-tree by `make` / __graft_entry__.build()).  Raises MtcError if it is
    absent -- there is deliberately no CPU fallback."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(_LIB_PATH):
        raise MtcError(f"{_LIB_PATH} not found: build it with `make` (hipcc --offload-arch=gfx950); "
                       "matryoshka_transcoder_amd has no CPU fallback")
    lib = C.CDLL(_LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().mtc_last_error()
        raise MtcError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")


def make_desc(B, A_s, A_t, D, prefix, dtype_code, unit_norm) -> Desc:
    if len(prefix) > MTC_MAX_PREFIX:
        raise MtcError(f"at most {MTC_MAX_PREFIX} prefix groups are supported, got {len(prefix)}")
    d = Desc()
    d.B, d.A_s, d.A_t, d.D, d.G = int(B), int(A_s), int(A_t), int(D), len(prefix)
    for i, p in enumerate(prefix):
        d.prefix[i] = int(p)
    d.dtype, d.unit_norm = int(dtype_code), int(bool(unit_norm))
    return d
