"""ctypes binding of libswr.so (include/swr.h) -- the only door to the device code.

There is NO CPU fallback: importing this module without the built library, or
launching with tensors that are not on a HIP device, raises.  PyTorch is used
for device memory and streams only (`tensor.data_ptr()`,
`torch.cuda.current_stream().cuda_stream`).
"""
import ctypes as C
import os

import torch

# SWR_LIB: another build of the same library (A/B timing of two kernel variants inside one process image)
_LIB_PATH = os.environ.get("SWR_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "_lib", "libswr.so")


class SwrError(RuntimeError):
    pass


def _load():
    if not os.path.exists(_LIB_PATH):
        raise ImportError(
            f"libswr.so not found at {_LIB_PATH}: build it with "
            "`python scenario-wise-rec_amd/build_native.py` (hipcc, gfx950). "
            "scenario_wise_rec has no CPU fallback.")
    return C.CDLL(_LIB_PATH)


# the ABI number of include/swr.h these bindings were written against (SWR_ABI_VERSION): argument lists changed between
# numbers, so a stale or variant libswr.so with another number would take shifted arguments -- refuse it
ABI_VERSION = 8

lib = _load()
lib.swr_abi_version.restype = C.c_int
if lib.swr_abi_version() != ABI_VERSION:
    raise ImportError(f"{_LIB_PATH} has ABI {lib.swr_abi_version()}, these bindings need {ABI_VERSION}: rebuild it with "
                      "`python scenario-wise-rec_amd/build_native.py --force`")

# ----------------------------------------------------------------------------- enums
I8, I16, I32, I64, U8, F16, BF16, F32, F64, BOOL = range(1, 11)
ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_SOFTMAX = 0, 1, 2, 3
FLAG_INDEX_OOR, FLAG_GRAD_RANGE = 1, 2

_DTYPES = {
    torch.int8: I8, torch.int16: I16, torch.int32: I32, torch.int64: I64, torch.uint8: U8,
    torch.float16: F16, torch.bfloat16: BF16, torch.float32: F32, torch.float64: F64, torch.bool: BOOL,
}
_ACTS = {None: ACT_NONE, "none": ACT_NONE, "relu": ACT_RELU, "sigmoid": ACT_SIGMOID, "softmax": ACT_SOFTMAX}


def dtype_code(t):
    try:
        return _DTYPES[t.dtype]
    except KeyError:
        raise SwrError(f"unsupported column dtype {t.dtype}")


# --------------------------------------------------------------------------- structs
class SparseSlot(C.Structure):
    _fields_ = [("weight", C.c_void_p), ("idx", C.c_void_p), ("vocab", C.c_int64), ("dim", C.c_int32),
                ("idx_dtype", C.c_int32), ("out_col", C.c_int32), ("hash_seed", C.c_uint32)]


class DenseSlot(C.Structure):
    _fields_ = [("values", C.c_void_p), ("dtype", C.c_int32), ("out_col", C.c_int32)]


class EmbedGradSlot(C.Structure):
    _fields_ = [("vocab", C.c_int64), ("dim", C.c_int32), ("in_col", C.c_int32), ("table_id", C.c_int32),
                ("mode", C.c_int32), ("grad_dense", C.c_void_p), ("urow", C.c_void_p), ("ugrad", C.c_void_p)]


class GemmArgs(C.Structure):
    _fields_ = [("M", C.c_int64), ("N", C.c_int32), ("K", C.c_int32),
                ("A", C.c_void_p), ("lda", C.c_int64), ("B", C.c_void_p), ("ldb", C.c_int64),
                ("bias", C.c_void_p), ("C", C.c_void_p), ("ldc", C.c_int64),
                ("a_scale", C.c_void_p), ("a_shift", C.c_void_p), ("a_relu", C.c_int32),
                ("accumulate", C.c_int32), ("stat_partials", C.c_void_p), ("groups", C.c_int32),
                ("gsA", C.c_int64), ("gsB", C.c_int64), ("gsC", C.c_int64), ("gsBias", C.c_int64),
                ("gsScale", C.c_int64), ("B_split", C.c_void_p), ("ld_split", C.c_int64), ("plane_stride", C.c_int64),
                ("n_compute", C.c_int32), ("a_exact_from", C.c_int32), ("c_act", C.c_int32), ("pad1", C.c_int32)]


class GemmTnArgs(C.Structure):
    _fields_ = [("M", C.c_int64), ("K1", C.c_int32), ("K2", C.c_int32),
                ("A", C.c_void_p), ("lda", C.c_int64), ("B", C.c_void_p), ("ldb", C.c_int64),
                ("C", C.c_void_p), ("ldc", C.c_int64), ("colsum", C.c_void_p),
                ("accumulate", C.c_int32), ("groups", C.c_int32),
                ("gsA", C.c_int64), ("gsB", C.c_int64), ("gsC", C.c_int64), ("gsColsum", C.c_int64),
                ("C2", C.c_void_p), ("ldc2", C.c_int64), ("c2_from", C.c_int64)]


class ActRange(C.Structure):
    _fields_ = [("col_lo", C.c_int32), ("col_hi", C.c_int32), ("act", C.c_int32), ("group", C.c_int32)]


MIX_MAX_OUT, MIX_MAX_SEL = 16, 16


class MixDesc(C.Structure):
    _fields_ = [("n_out", C.c_int32), ("n_sel", C.c_int32), ("H", C.c_int32), ("x_col", C.c_int32),
                ("g_col", C.c_int32), ("g_stride", C.c_int32), ("sel", (C.c_uint8 * MIX_MAX_SEL) * MIX_MAX_OUT)]


class BnMixArgs(C.Structure):
    _fields_ = [("M", C.c_int64), ("ne", C.c_int32), ("H", C.c_int32), ("D", C.c_int32), ("pad", C.c_int32),
                ("Z", C.c_void_p), ("ldz", C.c_int64), ("scale", C.c_void_p), ("shift", C.c_void_p),
                ("P", C.c_void_p), ("ldp", C.c_int64), ("dP", C.c_void_p), ("lddp", C.c_int64),
                ("mean", C.c_void_p), ("rstd", C.c_void_p), ("dY", C.c_void_p), ("lddy", C.c_int64),
                ("bn_partials", C.c_void_p), ("G", C.c_void_p)]


class TowerArgs(C.Structure):
    _fields_ = [("M", C.c_int64), ("G", C.c_int32), ("K", C.c_int32), ("H", C.c_int32), ("accumulate", C.c_int32),
                ("X", C.c_void_p), ("ldx", C.c_int64), ("W1", C.c_void_p), ("b1", C.c_void_p),
                ("Z1", C.c_void_p), ("ldz", C.c_int64), ("stat_partials", C.c_void_p),
                ("scale", C.c_void_p), ("shift", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p),
                ("gamma", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p), ("V", C.c_void_p), ("ldv", C.c_int64),
                ("dV", C.c_void_p), ("lddv", C.c_int64), ("ca", C.c_void_p), ("cb", C.c_void_p), ("cc", C.c_void_p),
                ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("dw2", C.c_void_p), ("db2", C.c_void_p),
                ("dZ1", C.c_void_p), ("lddz", C.c_int64), ("dX", C.c_void_p), ("lddx", C.c_int64),
                ("sel_domain", C.c_void_p), ("sel_y", C.c_void_p), ("sel_p", C.c_void_p), ("sel_dloss", C.c_void_p),
                ("sel_dom_dtype", C.c_int32), ("sel_y_dtype", C.c_int32)]


_P8 = C.c_void_p * 8


class StarLayerArgs(C.Structure):
    _fields_ = [("D", C.c_int32), ("in_dim", C.c_int32), ("out_dim", C.c_int32), ("first", C.c_int32),
                ("Ws", C.c_void_p), ("bs", C.c_void_p), ("Wd", _P8), ("bd", _P8),
                ("gamma_s", C.c_void_p), ("beta_s", C.c_void_p), ("gamma_d", _P8), ("beta_d", _P8),
                ("W_eff", _P8), ("b_eff", _P8), ("dW_eff", _P8), ("db_eff", _P8),
                ("dWs", C.c_void_p), ("dbs", C.c_void_p), ("dWd", _P8), ("dbd", _P8),
                ("dgamma_s", C.c_void_p), ("dbeta_s", C.c_void_p), ("dgamma_d", _P8), ("dbeta_d", _P8),
                ("accumulate", C.c_int32), ("pad", C.c_int32)]


class TakeColumn(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("elem_bytes", C.c_int32), ("pad", C.c_int32)]


class DpTable(C.Structure):
    _fields_ = [("row_off", C.c_int64), ("grad_off", C.c_int64), ("n", C.c_int64), ("dim", C.c_int32), ("pad", C.c_int32),
                ("out_row", C.c_void_p), ("out_grad", C.c_void_p)]


class LayerNormArgs(C.Structure):
    _fields_ = [("M", C.c_int64), ("G", C.c_int32), ("N", C.c_int32), ("relu", C.c_int32), ("accumulate", C.c_int32),
                ("eps", C.c_float), ("pad", C.c_int32), ("X", C.c_void_p), ("ldx", C.c_int64), ("gamma", C.c_void_p),
                ("beta", C.c_void_p), ("Y", C.c_void_p), ("ldy", C.c_int64), ("mean", C.c_void_p), ("rstd", C.c_void_p),
                ("dY", C.c_void_p), ("lddy", C.c_int64), ("dX", C.c_void_p), ("lddx", C.c_int64), ("dgamma", C.c_void_p),
                ("dbeta", C.c_void_p)]


ADAM_MAX_TABLES = 16        # SWR_ADAM_MAX_TABLES


class AdamTable(C.Structure):
    _fields_ = [("p", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("last", C.c_void_p), ("claim", C.c_void_p),
                ("vocab", C.c_int64), ("dim", C.c_int32), ("idx_dtype", C.c_int32), ("idx", C.c_void_p),
                ("hash_seed", C.c_uint32), ("pad", C.c_uint32), ("n", C.c_int64), ("from_step", C.c_void_p), ("rows", C.c_void_p),
                ("urow", C.c_void_p), ("ugrad", C.c_void_p)]


class OnehotTable(C.Structure):
    _fields_ = [("grad", C.c_void_p), ("vocab", C.c_int32), ("dim", C.c_int32), ("oh_off", C.c_int32), ("w_col", C.c_int32)]


class FlPiece(C.Structure):
    _fields_ = [("kind", C.c_int32), ("slot", C.c_int32), ("off", C.c_int32), ("n_valid", C.c_int32), ("w_col", C.c_int32),
                ("pad", C.c_int32)]


class FlPlan(C.Structure):
    _fields_ = [("sparse_host", C.c_void_p), ("n_sparse", C.c_int32), ("dense_host", C.c_void_p), ("n_dense", C.c_int32),
                ("n_keys", C.c_int32), ("n_real_groups", C.c_int32), ("piece", FlPiece * 32), ("oh_width", C.c_int32),
                ("oh_off_host", C.c_void_p), ("B", C.c_int64), ("N", C.c_int32), ("pad", C.c_int32)]


class FlOffsets(C.Structure):
    _fields_ = [("zero", C.c_int64), ("planes", C.c_int64), ("a3f", C.c_int64), ("b3", C.c_int64), ("keys", C.c_int64),
                ("mask", C.c_int64), ("mask_t", C.c_int64), ("voff", C.c_int64), ("densef", C.c_int64), ("b3x", C.c_int64), ("total", C.c_int64),
                ("nd4", C.c_int32), ("n_fpieces", C.c_int32)]


FL_ZERO, FL_PLANES, FL_ROWS, FL_DENSE = 0, 1, 2, 3


class AdamHyper(C.Structure):
    _fields_ = [("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("weight_decay", C.c_double), ("step", C.c_int64), ("step_size", C.c_float),
                ("inv_bc2_sqrt", C.c_float), ("one_minus_b1", C.c_float), ("b2", C.c_float),
                ("one_minus_b2", C.c_float), ("eps_f", C.c_float), ("wd_f", C.c_float), ("hist_mask", C.c_uint32)]


# ------------------------------------------------------------------------ signatures
_P, _I, _L, _F, _Z = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t
_SIGS = {
    "swr_abi_version": (C.c_int, []),
    "swr_spin_us": (C.c_int, [C.c_int, _P]),
    "swr_stamp": (C.c_int, [_P, _P]),
    "swr_tower_dw_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "swr_tower_dw_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int, C.c_int, C.c_int]),
    "swr_tower_dw": (C.c_int, [_P, C.c_int64, _P, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _P, C.c_size_t, _P]),
    "swr_zero": (C.c_int, [_P, _Z, _P]),
    "swr_status_str": (C.c_char_p, [_I]),
    "swr_device_available": (C.c_int, []),
    "swr_embed_gather_fwd": (C.c_int, [_P, _I, _P, _I, _L, _P, _L, _P, _P, _P]),
    "swr_embed_gather_fwd_onehot": (C.c_int, [_P, _I, _P, _I, _L, _P, _L, _P, _P, _I, _I, _I, _P, _P]),
    "swr_fold_first_layer_fwd": (C.c_int, [_P, _L, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _L, _P, _I, _P, _L, _P]),
    "swr_fold_first_layer_bwd": (C.c_int, [_P, _L, _P, _I, _I, _I, _I, _P, _P, _P, _I, _P, _L, _P, _I, _P]),
    "swr_fold_first_layer_bwd_tables": (C.c_int, [_P, _L, _P, _I, _I, _I, _I, _P, _P, _P, _I, _P, _L, _P, _I, _P, _L, _P, _I, _P]),
    "swr_onehot_table_grads": (C.c_int, [_P, _L, _P, _L, _I, _P, _I, _I, _P]),
    "swr_fl_layout": (C.c_int, [_P, _P]),
    "swr_fl_workspace_bytes": (_Z, [_P]),
    "swr_fl_prep": (C.c_int, [_P, _P, _L, _I, _P, _P, _I, _P, _I, _P, _L, _P, _P]),
    "swr_fl_keys": (C.c_int, [_P, _P, _P, _P]),
    "swr_fl_fwd": (C.c_int, [_P, _P, _P, _P, _L, _P, _P]),
    "swr_bn_bwd_dx_supported": (C.c_int, [_I, _I]),
    "swr_bn_bwd_dx": (C.c_int, [_P, _P, _P, _L, _P, _L, _P, _P, _P, _P, _I, _P, _L, _P, _L, _P]),
    "swr_fl_dw_supported": (C.c_int, [_P, _L]),
    "swr_fl_dw_workspace_bytes": (_Z, [_P]),
    "swr_fl_dw": (C.c_int, [_P, _P, _P, _L, _P, _L, _P, _P, _Z, _P]),
    "swr_fl_dw_bn_supported": (C.c_int, [_P, _L, _L]),
    "swr_dw_tr_mode": (C.c_int, [_I]),
    "swr_transpose_groups": (C.c_int, [_P, _I, _I, _I, _P, _P]),
    "swr_fl_dw_bn": (C.c_int, [_P, _P, _P, _L, _P, _L, _P, _P, _P, _P, _P, _L, _P, _P, _Z, _P]),
    "swr_adam_catchup_multi": (C.c_int, [_P, _I, _P, _P, _P]),
    "swr_adam_rows_multi": (C.c_int, [_P, _I, _P, _P]),
    "swr_embed_bag_fwd": (C.c_int, [_P, _L, _I, _P, _I, _L, _I, _I, _I, _L, C.c_uint32, _P, _L, _I, _P, _P, _P, _P]),
    "swr_embed_bag_bwd_expand": (C.c_int, [_P, _L, _I, _I, _I, _I, _P, _L, _P, _P]),
    "swr_embed_bwd_workspace_bytes": (_Z, [_P, _I, _L]),
    "swr_embed_bwd": (C.c_int, [_P, _I, _P, _P, _L, _L, _P, _Z, _P, _P]),
    "swr_embed_bwd_sort": (C.c_int, [_P, _I, _P, _L, _P, _Z, _P]),
    "swr_embed_bwd_reduce": (C.c_int, [_P, _I, _P, _P, _L, _L, _P, _Z, _P, _P]),
    "swr_embed_bwd_reduce_part": (C.c_int, [_P, _I, _P, _P, _L, _L, _I, _P, _Z, _P, _P]),
    "swr_gemm_nt": (C.c_int, [_P, _P]),
    "swr_gemm_nn": (C.c_int, [_P, _P]),
    "swr_split_ld": (C.c_int64, [_L]),
    "swr_gemm_precision_mode": (C.c_int, []),
    "swr_split_weights": (C.c_int, [_P, _L, _I, _I, _P, _P, _P]),
    "swr_gemm_tn_workspace_bytes": (_Z, [_P]),
    "swr_gemm_tn": (C.c_int, [_P, _P, _Z, _P]),
    "swr_bn_finalize": (C.c_int, [_P, _I, _L, _I, _P, _P, _F, _F, _P, _P, _P, _I, _P, _P, _P, _P, _P]),
    "swr_col_moments": (C.c_int, [_P, _L, _L, _I, _P, _P]),
    "swr_bn_eval_coeffs": (C.c_int, [_P, _P, _P, _P, _F, _I, _P, _P, _P]),
    "swr_affine_act_fwd": (C.c_int, [_P, _L, _P, _P, _P, _I, _P, _L, _L, _I, _P]),
    "swr_bn_act_bwd_stats": (C.c_int, [_P, _L, _P, _L, _P, _L, _P, _P, _P, _I, _P, _L, _I, _P]),
    "swr_bn_bwd_finalize": (C.c_int, [_P, _I, _L, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P]),
    "swr_act_bwd_apply": (C.c_int, [_P, _L, _P, _L, _P, _L, _P, _P, _P, _P, _P, _I, _P, _L, _L, _I, _P]),
    "swr_moe_mix_fwd": (C.c_int, [_P, _P, _L, _P, _L, _P, _L, _L, _P]),
    "swr_moe_mix_bwd": (C.c_int, [_P, _P, _L, _P, _L, _P, _L, _P, _L, _P, _L, _I, _L, _P]),
    "swr_select_fwd": (C.c_int, [_P, _L, _I, _P, _I, _I, _P, _P, _L, _P]),
    "swr_select_bwd": (C.c_int, [_P, _P, _I, _P, _I, _I, _I, _P, _L, _P, _L, _P]),
    "swr_bce_workspace_bytes": (_Z, [_L]),
    "swr_bce_fwd": (C.c_int, [_P, _P, _I, _L, _P, _P, _Z, _P]),
    "swr_bce_bwd": (C.c_int, [_P, _P, _I, _L, _P, _P, _P]),
    "swr_select_bce_fwd": (C.c_int, [_P, _L, _I, _P, _I, _P, _I, _L, _P, _P, _P, _Z, _P, _P]),
    "swr_select_bce_fwd_adv": (C.c_int, [_P, _L, _I, _P, _I, _P, _I, _L, _P, _P, _P, _Z, _P, _P, _P, _L, _P]),
    "swr_select_bce_bwd": (C.c_int, [_P, _P, _I, _I, _P, _I, _L, _P, _P, _L, _P]),
    "swr_tower_head_select_bce_fwd": (C.c_int, [_P, _L, _I, _I, _P, _P, _P, _P, _P, _I, _P, _I, _L, _P, _P, _P, _Z, _P, _P]),
    "swr_tower_head_select_bce_fwd_adv": (C.c_int, [_P, _L, _I, _I, _P, _P, _P, _P, _P, _I, _P, _I, _L, _P, _P, _P, _Z, _P, _P, _P, _L, _P]),
    "swr_bnmix_supported": (C.c_int, [_I, _I, _I]),
    "swr_bnmix_tile_rows": (C.c_int, []),
    "swr_bnmix_fwd": (C.c_int, [_P, _P]),
    "swr_bnmix_bwd": (C.c_int, [_P, _P]),
    "swr_tower_supported": (C.c_int, [_I, _I]),
    "swr_tower_fwd_linear": (C.c_int, [_P, _P]),
    "swr_tower_fwd_head": (C.c_int, [_P, _P]),
    "swr_tower_bwd_workspace_bytes": (_Z, [_L, _I, _I]),
    "swr_tower_bwd": (C.c_int, [_P, _P, _Z, _P]),
    "swr_rowmat_fwd": (C.c_int, [_P, _P, _P, _L, _I, _I, _P]),
    "swr_rowmat_bwd": (C.c_int, [_P, _P, _P, _P, _P, _I, _L, _I, _I, _P]),
    "swr_mul_fwd": (C.c_int, [_P, _P, _P, _L, _P]),
    "swr_mul_scale_fwd": (C.c_int, [_P, _P, _F, _P, _L, _P]),
    "swr_mul_scale_bwd": (C.c_int, [_P, _P, _P, _F, _P, _P, _L, _P]),
    "swr_mul_sigmoid_fwd": (C.c_int, [_P, _P, _F, _P, _L, _P]),
    "swr_mul_sigmoid_bwd": (C.c_int, [_P, _P, _P, _F, _P, _P, _L, _P]),
    "swr_add_fwd": (C.c_int, [_P, _P, _P, _L, _P]),
    "swr_colsum_workspace_bytes": (_Z, [_L, _I]),
    "swr_colsum": (C.c_int, [_P, _L, _L, _I, _P, _I, _P, _Z, _P]),
    "swr_adam_advance": (C.c_int, [_P, _P, _L, _P]),
    "swr_adam_dense": (C.c_int, [_P, _P, _P, _P, _L, _I, _P, _P]),
    "swr_adam_rows": (C.c_int, [_P, _P, _P, _L, _I, _P, _P, _L, _P, _P, _P, _P]),
    "swr_adam_dense_rows": (C.c_int, [_P, _P, _P, _P, _L, _I, _P, _P, _P, _L, _I, _P, _P, _L, _P, _P, _P]),
    "swr_adam_catchup_rows": (C.c_int, [_P, _P, _P, _P, _P, _L, _I, _P, _I, C.c_uint32, _L, _P, _P, _P, _Z, _P]),
    "swr_adam_flush": (C.c_int, [_P, _P, _P, _P, _L, _I, _P, _P, _P]),
    "swr_adam_sweep_untouched": (C.c_int, [_P, _P, _P, _L, _I, _P, _I, _P, _P]),
    "swr_routed_mmoe_eval_supported": (C.c_int, [_I, _I, _I, _I]),
    "swr_routed_mmoe_eval": (C.c_int, [_P, _L, _L, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    "swr_take_rows": (C.c_int, [_P, _I, _P, _L, _L, _P, _P]),
    "swr_layernorm_fwd": (C.c_int, [_P, _P]),
    "swr_layernorm_bwd_workspace_bytes": (_Z, [_L, _I, _I]),
    "swr_layernorm_bwd": (C.c_int, [_P, _P, _Z, _P]),
    "swr_block_select_fwd": (C.c_int, [_P, _L, _P, _I, _I, _I, _L, _P, _L, _P]),
    "swr_block_select_bwd": (C.c_int, [_P, _L, _P, _I, _I, _I, _L, _P, _L, _P]),
    "swr_star_layer_fwd": (C.c_int, [_P, _P]),
    "swr_star_layer_bwd": (C.c_int, [_P, _P]),
    "swr_star_layers_fwd": (C.c_int, [_P, _I, _P]),
    "swr_star_layers_bwd": (C.c_int, [_P, _I, _P]),
    "swr_eval_metrics_workspace_bytes": (_Z, [_L, _I]),
    "swr_eval_metrics": (C.c_int, [_P, _P, _I, _P, _I, _L, _I, _P, _P, _P, _Z, _P]),
    "swr_dp_finish": (C.c_int, [_P, _L, _L, _P, _P, _L, _P, _I, _I, _F, _P]),
}
EXPORTS = tuple(_SIGS)
for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)          # AttributeError here = the library does not match include/swr.h
    _fn.restype, _fn.argtypes = _res, _args


def check(rc, what):
    if rc != 0:
        raise SwrError(f"{what}: {lib.swr_status_str(rc).decode()} ({rc})")


# --------------------------------------------------------------------------- helpers
def require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise SwrError("scenario_wise_rec (MI355X build) runs on HIP device tensors only; got a "
                           f"{t.device} tensor. There is no CPU fallback.")


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def f32c(t):
    """fp32, unit stride in the last dim (rows may be strided)."""
    if t.dtype != torch.float32:
        t = t.float()
    if t.dim() >= 1 and t.stride(-1) != 1:
        t = t.contiguous()
    if t.dim() == 2 and t.stride(0) < t.shape[1]:
        t = t.contiguous()
    return t


_err_flags = {}


def err_flag(device):
    """Per-device sticky error word written by the kernels (index out of range, gradient range)."""
    key = torch.device(device).index or 0
    if key not in _err_flags:
        _err_flags[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _err_flags[key]


_tickets = {}


def ticket(device):
    """A zeroed device word per device for kernels that elect their last workgroup (they leave it zero).  Users are
    serialised on one stream (the loss of a step); allocate before any graph capture (the warm-up steps do)."""
    key = torch.device(device).index or 0
    if key not in _tickets:
        _tickets[key] = torch.zeros(4, dtype=torch.int32, device=device)
    return _tickets[key]


def check_errors(device=None):
    """Synchronising check of the device error word; raises like the reference would."""
    for key, flag in list(_err_flags.items()):
        v = int(flag.item())
        if v:
            flag.zero_()
            if v & FLAG_INDEX_OOR:
                raise IndexError("index out of range in self")     # torch's nn.Embedding message
            if v & FLAG_GRAD_RANGE:
                raise SwrError("embedding gradient outside the fixed-point accumulator range (|g| >= 2^20)")


def act_ranges(acts, n_cols):
    """acts: None | str | list of (lo, hi, act, group)."""
    if acts is None or isinstance(acts, str):
        acts = [(0, n_cols, acts, 1)]
    arr = (ActRange * len(acts))()
    for i, (lo, hi, a, g) in enumerate(acts):
        arr[i] = ActRange(lo, hi, _ACTS[a.lower() if isinstance(a, str) else a], g)
    return arr, len(acts)
