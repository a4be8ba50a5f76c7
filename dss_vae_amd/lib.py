"""ctypes binding of the C-ABI in include/dssvae_hip.h (libdssvae_hip.so, built in-tree by
``__graft_entry__.build()`` / ``make -C dss_vae_amd/csrc``).

There is no fallback: if the library is missing, loading raises, and every op in
``dss_vae_amd.ops`` refuses non-CUDA tensors.  PyTorch is used only for device memory and the
current HIP stream; tensors cross the boundary as raw device pointers.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DSSVAE_LIB") or os.path.join(_HERE, "lib", "libdssvae_hip.so")  # env: development builds

ABI_VERSION = 9  # include/dssvae_hip.h: DSSVAE_ABI_VERSION
GRU_MAX_INST = 8
E_TOOBIG = -2
_ERRORS = {-1: "DSSVAE_EINVAL (rejected argument)", -2: "DSSVAE_ETOOBIG (persistent grid not co-resident)",
           -3: "DSSVAE_ETIMEOUT (persistent kernel gave up waiting)"}


class GruFwdInst(C.Structure):
    _fields_ = [("xproj", C.c_void_p), ("ldx", C.c_int64), ("whh", C.c_void_p), ("bhh", C.c_void_p),
                ("h0", C.c_void_p), ("lengths", C.c_void_p), ("out", C.c_void_p), ("ldo", C.c_int64),
                ("h_last", C.c_void_p), ("gates", C.c_void_p), ("B", C.c_int32), ("T", C.c_int32),
                ("H", C.c_int32), ("reverse", C.c_int32), ("t0", C.c_int32), ("T_total", C.c_int32),
                ("rows", C.c_int32), ("cell", C.c_int32), ("c0", C.c_void_p), ("c_last", C.c_void_p)]


class GruBwdInst(C.Structure):
    _fields_ = [("whh", C.c_void_p), ("h0", C.c_void_p), ("lengths", C.c_void_p), ("out", C.c_void_p),
                ("ldo", C.c_int64), ("gates", C.c_void_p), ("dout", C.c_void_p), ("ldd", C.c_int64),
                ("dh_last", C.c_void_p), ("dgx", C.c_void_p), ("dgh", C.c_void_p), ("dh0", C.c_void_p),
                ("B", C.c_int32), ("T", C.c_int32), ("H", C.c_int32), ("reverse", C.c_int32),
                ("dbias_x", C.c_void_p), ("dbias_h", C.c_void_p), ("t0", C.c_int32), ("T_total", C.c_int32),
                ("rows", C.c_int32), ("cell", C.c_int32), ("dc_last", C.c_void_p), ("dc0", C.c_void_p)]


CELL_GRU, CELL_LSTM = 0, 1


GEMM_MAX_GROUP = 8
SUMSQ_PARTIALS = 2048
STEP_WORDS = 128  # DSSVAE_STEP_WORDS


class GemmProblem(C.Structure):
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int64), ("B", C.c_void_p), ("ldb", C.c_int64), ("C", C.c_void_p),
                ("ldc", C.c_int64), ("bias", C.c_void_p), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32)]


DECODE_MAX_LAYERS = 4
DECODE_NOALLOC_BYTES = 16 << 20


class DecodeParams(C.Structure):
    _fields_ = [("embedding", C.c_void_p), ("ld_embedding", C.c_int64), ("n_embeddings", C.c_int64),
                ("w_ih", C.c_void_p * DECODE_MAX_LAYERS), ("w_hh", C.c_void_p * DECODE_MAX_LAYERS),
                ("b_ih", C.c_void_p * DECODE_MAX_LAYERS), ("b_hh", C.c_void_p * DECODE_MAX_LAYERS),
                ("w_out", C.c_void_p), ("ld_out", C.c_int64), ("b_out", C.c_void_p),
                ("E", C.c_int32), ("H", C.c_int32), ("V", C.c_int32), ("n_layers", C.c_int32)]


_P, _I, _L, _F, _U64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint64
_SIGNATURES = {
    "dssvae_abi_version": ([], _I),
    "dssvae_device_cus": ([C.POINTER(C.c_int)], _I),
    "dssvae_gemm_last_launches": ([], _I),
    "dssvae_gemm_f32": ([_P, _L, _I, _P, _P, _L, _I, _P, _L, _P, _I, _I, _I, _F, _F, _P], _I),
    "dssvae_gemm_f32_splitk": ([_P, _L, _I, _P, _P, _L, _I, _P, _L, _P, _I, _I, _I, _F, _F, _I, _P], _I),
    "dssvae_gemm_f32_grouped": ([C.POINTER(GemmProblem), _I, _I, _I, _P], _I),
    "dssvae_linear_lse_fwd": ([_P, _L, _I, _P, _L, _I, _P, _L, _P, _I, _I, _I, _P, _P, _P, _I, _P, _L, _P], _I),
    "dssvae_softmax_grad_inplace": ([_P, _L, _P, _P, _I, _I, _P, _P], _I),
    "dssvae_ce_rowscale": ([_P, _P, _I, _I, _P, _L, _P], _I),
    "dssvae_linear_lse_fwd_norm": ([_P, _L, _I, _P, _L, _I, _P, _L, _P, _I, _I, _P, _L, _P, _P, _L, _P], _I),
    "dssvae_lse_combine_tiles": ([_P, _P, _I, _I, _L, _P, _P], _I),
    "dssvae_ce_onehot_fixup": ([_P, _L, _P, _I, _P, _L, _I, _P, _L, _P, _L, _I, _P, _L, _P], _I),
    "dssvae_ce_wgrad_cols": ([_P, _L, _I, _I, _P, _L, _P, _L, _I, _P, _L, _F, _P], _I),
    "dssvae_gemm_f32_dlogits": ([_P, _L, _I, _I, _P, _L, _I, _P, _L, _P, _L, _I, _F, _P], _I),
    "dssvae_lse_combine": ([_P, _P, _P, _I, _I, _P, _P, _P], _I),
    "dssvae_picked_logit": ([_P, _L, _I, _P, _L, _I, _P, _L, _P, _P, _I, _L, _I, _P, _P], _I),
    "dssvae_linear_ce_dlogits": ([_P, _L, _I, _P, _L, _I, _P, _L, _P, _P, _P, _I, _I, _P, _L, _P, _P], _I),
    "dssvae_onehot_sub": ([_P, _L, _P, _I, _L, _P, _I, _L, _L, _P, _P], _I),
    "dssvae_colsum": ([_P, _L, _I, _I, _P, _I, _P], _I),
    "dssvae_embed_gather": ([_P, _L, _P, _I, _I, _P, _P, _L, _P], _I),
    "dssvae_embed_scatter_add": ([_P, _L, _P, _I, _I, _P, _P, _L, _P], _I),
    "dssvae_dropout_mask": ([_P, _L, _F, _U64, _P], _I),
    "dssvae_dropout_apply": ([_P, _L, _F, _U64, _P, _P], _I),
    "dssvae_ce_count": ([_P, _I, _L, _I, _P, _P], _I),
    "dssvae_ce_finish": ([_P, _L, _I, _P, _L, _I, _P, _L, _P, _P, _I, _L, _P, _I, _I, _P, _P, _P], _I),
    "dssvae_unk_replace": ([_P, _L, _P, _F, _U64, _L, _L, _L, _L, _P, _P], _I),
    "dssvae_reparam_kl_fwd": ([_P, _P, _P, _L, _P, _P, _P], _I),
    "dssvae_reparam_kl_bwd": ([_P, _P, _P, _P, _P, _L, _P, _P, _P], _I),
    "dssvae_weighted_sums_fwd": ([C.POINTER(C.c_void_p), C.POINTER(C.c_int32), _I, C.POINTER(C.c_float), _I, _P, _P], _I),
    "dssvae_weighted_sums_bwd": ([C.POINTER(C.c_void_p), C.POINTER(C.c_int32), _I, C.POINTER(C.c_float), _I, C.POINTER(C.c_void_p), _P], _I),
    "dssvae_gru_workspace_bytes": ([_I, C.POINTER(C.c_int32), C.POINTER(C.c_int32)], _L),
    "dssvae_gru_resident_workgroups": ([_I, _I, _I, C.POINTER(C.c_int)], _I),
    "dssvae_gru_layout": ([_I, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), _I, _I,
                           C.POINTER(C.c_int)], _I),
    "dssvae_gru_fwd": ([C.POINTER(GruFwdInst), _I, _P, _L, _P], _I),
    "dssvae_gru_bwd": ([C.POINTER(GruBwdInst), _I, _P, _L, _P], _I),
    "dssvae_gru_check": ([_P, _P], _I),
    "dssvae_gru_error_word": ([_P], _I),
    "dssvae_hold_cus": ([_P, _I, _I, _I, _P, _P], _I),
    "dssvae_gru_cell": ([_P, _P, _P, _I, _I, _P, _P], _I),
    "dssvae_gru_step_fwd": ([_P, _L, _P, _P, _P, _P, _I, _I, _I, _P, _P, _L, _P, _L, _P], _I),
    "dssvae_gru_step_bwd": ([_P, _L, _P, _L, _P, _P, _I, _I, _I, _P, _P, _L, _P, _P], _I),
    "dssvae_lstm_step_fwd": ([_P, _L, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _L, _P, _L, _P], _I),
    "dssvae_lstm_step_bwd": ([_P, _L, _P, _L, _P, _P, _P, _I, _I, _I, _P, _L, _P, _P, _P], _I),
    "dssvae_skinny_supported": ([_I, _I, _I, _L, _L, _L], _I),
    "dssvae_skinny_argmax_splits": ([_I, _I], _I),
    "dssvae_skinny_argmax_rows": ([_I, _I], _I),
    "dssvae_skinny_argmax_padded_rows": ([_I, _I], _I),
    "dssvae_skinny_linear": ([_P, _L, _I, _P, _L, _I, _P, _L, _P, _I, _I, _P, _L, _P], _I),
    "dssvae_skinny_gru_step": ([_P, _L, _I, _P, _L, _I, _P, _L, _P, _L, _P, _P, _I, _P, _L, _P], _I),
    "dssvae_skinny_gru_step_pick": ([_P, _P, _I, _P, _P, _L, _I, _P, _L, _P, _L, _I, _P, _L, _P, _L, _P, _P, _I, _P, _L,
                                    _P], _I),
    "dssvae_skinny_argmax": ([_P, _L, _I, _P, _L, _I, _P, _L, _P, _I, _I, _P, _P, _P, _P], _I),
    "dssvae_greedy_decode_workspace_bytes": ([_I, _I, _I, _I, _I, _I], _L),
    "dssvae_greedy_decode_lstm_workspace_bytes": ([_I, _I, _I, _I, _I, _I], _L),
    "dssvae_greedy_decode_lstm": ([C.POINTER(DecodeParams), _P, _P, _I, _I, _L, _I, _P, _P, _P, _P, _L, _P], _I),
    "dssvae_skinny_lstm_step": ([_P, _L, _I, _P, _L, _I, _P, _P, _L, _P, _L, _P, _L, _P, _P, _I, _P, _L, _P], _I),
    "dssvae_skinny_lstm_step_pick": ([_P, _P, _I, _P, _P, _L, _I, _P, _L, _P, _L, _I, _P, _P, _L, _P, _L, _P, _L, _P, _P,
                                      _I, _P, _L, _P], _I),
    "dssvae_greedy_decode": ([C.POINTER(DecodeParams), _P, _I, _I, _L, _I, _P, _P, _P, _L, _P], _I),
    "dssvae_greedy_decode_noalloc": ([C.POINTER(DecodeParams), _P, _I, _I, _L, _I, _P, _P, _P], _I),
    "dssvae_decode_profile": ([_I], _I),
    "dssvae_decode_profile_read": ([C.POINTER(C.c_double), C.POINTER(C.c_int)], _I),
    "dssvae_sumsq": ([_P, _L, _P, _P, _P], _I),
    "dssvae_adam_step": ([_P, _P, _P, _P, _L, _F, _F, _F, _F, _I, _P, _F, _P, _P, _P], _I),
    "dssvae_set_words": ([_P, C.POINTER(C.c_uint32), _I, _P], _I),
    "dssvae_dropout_mask_dyn": ([_P, _L, _F, _U64, _P, _P], _I),
    "dssvae_dropout_apply_dyn": ([_P, _L, _F, _U64, _P, _P, _P], _I),
    "dssvae_unk_replace_dyn": ([_P, _L, _P, _F, _P, _U64, _P, _L, _L, _L, _L, _P, _P], _I),
    "dssvae_weighted_sums_fwd_dyn": ([C.POINTER(C.c_void_p), C.POINTER(C.c_int32), _I, C.POINTER(C.c_float),
                                      C.POINTER(C.c_float), _P, _I, _P, _P], _I),
    "dssvae_weighted_sums_bwd_dyn": ([C.POINTER(C.c_void_p), C.POINTER(C.c_int32), _I, C.POINTER(C.c_float),
                                      C.POINTER(C.c_float), _P, _I, C.POINTER(C.c_void_p), _P], _I),
    "dssvae_adam_step_dyn": ([_P, _P, _P, _P, _L, _P, _F, _F, _F, _P, _F, _P, _P], _I),
    "dssvae_batch_rows": ([_P, _P, _P, _I, _I, _I, _L, _L, _L, _P, _P, _P], _I),
    "dssvae_keep_rows": ([_P, _I, _I, _P, _I, _P], _I),
}
EXPORTS = tuple(_SIGNATURES)

_lib = None


def load():
    """Load the HIP library once; fail loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "dss_vae_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C dss_vae_amd/csrc`; there is no CPU fallback" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (argtypes, restype) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes, fn.restype = argtypes, restype
        if lib.dssvae_abi_version() != ABI_VERSION:
            raise RuntimeError("dss_vae_amd: %s is ABI version %d, this package binds version %d -- rebuild it "
                               "(make -C dss_vae_amd/csrc)" % (LIB_PATH, lib.dssvae_abi_version(), ABI_VERSION))
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s" % (what, _ERRORS.get(rc, "hipError %d" % rc)))


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("dss_vae_amd ops run on the MI355X only (got a %s tensor); there is no CPU path"
                               % t.device)
