"""ctypes binding of libmtlora_hip.so (include/mtlora_hip.h).

The library is the ONLY compute path: if it cannot be loaded, or a tensor is not on a GPU, the
callers raise -- there is no eager / CPU fallback (the oracle under oracle/ is test-only).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_uint64, c_void_p

import torch

MAX_TASKS = 8
ABI_VERSION = 8
F32, BF16, F16 = 0, 1, 2
_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmtlora_hip.so")


class LinearDesc(Structure):
    _fields_ = [("M", c_int64), ("K", c_int64), ("N", c_int64), ("dtype", c_int32), ("mode", c_int32),
                ("T", c_int32), ("r_s", c_int32), ("r_t", c_int32 * MAX_TASKS), ("scale_s", c_float),
                ("scale_t", c_float * MAX_TASKS), ("has_x_tasks", c_int32), ("dropout_p", c_float),
                ("seed", c_uint64), ("seed_offset", c_void_p), ("bwd_phase", c_int32),
                # kernel selection (ABI v6): 0 = the library's own choice; see include/mtlora_hip.h
                ("sel_stream", c_int32), ("sel_dense", c_int32), ("sel_tn", c_int32), ("sel_projk", c_int32),
                ("max_cu", c_int32), ("packed", c_void_p),
                # ABI v8: role inside an Mlp with implicit task hidden tensors (HID_* flags) and its extra pointer
                ("hid", c_int32), ("hid_pad_", c_int32), ("hid_ptr", c_void_p)]


HID_FWD_BASE, HID_P_GIVEN, HID_Q_GIVEN = 1, 2, 4


class AttnDesc(Structure):
    _fields_ = [("B", c_int64), ("H", c_int32), ("W", c_int32), ("window_size", c_int32), ("shift", c_int32),
                ("num_heads", c_int32), ("head_dim", c_int32), ("image_layout", c_int32), ("dtype", c_int32),
                ("scale", c_float), ("mask_value", c_float)]


class BlockDesc(Structure):
    """mtlora_block_desc (ABI v7): one SwinTransformerBlock without task outputs"""
    _fields_ = [("B", c_int64), ("H", c_int32), ("W", c_int32), ("C", c_int32), ("hidden", c_int32), ("num_heads", c_int32),
                ("window_size", c_int32), ("shift", c_int32), ("dtype", c_int32), ("x_dtype", c_int32), ("has_norm1", c_int32),
                ("eps1", c_float), ("eps2", c_float), ("eps_next", c_float), ("attn_scale", c_float), ("mask_value", c_float),
                ("lin", LinearDesc * 4)]


class BlockParams(Structure):
    _fields_ = [("norm1_g", c_void_p), ("norm1_b", c_void_p), ("norm2_g", c_void_p), ("norm2_b", c_void_p), ("next_g", c_void_p),
                ("next_b", c_void_p), ("W", c_void_p * 4), ("Wt", c_void_p * 4), ("bias", c_void_p * 4), ("A", c_void_p * 4),
                ("Bf", c_void_p * 4), ("attn_bias", c_void_p), ("mask_ids", c_void_p), ("mask", c_void_p), ("scale1", c_void_p),
                ("scale2", c_void_p)]


class BlockGrads(Structure):
    _fields_ = [("g_x", c_void_p), ("g_normed", c_void_p), ("d_norm1_g", c_void_p), ("d_norm1_b", c_void_p), ("d_norm2_g", c_void_p),
                ("d_norm2_b", c_void_p), ("d_next_g", c_void_p), ("d_next_b", c_void_p), ("dA", c_void_p * 4), ("dB", c_void_p * 4),
                ("dbias", c_void_p)]


PROF_KINDS = 24


class ProfSummary(Structure):
    _fields_ = [("count", c_int64 * PROF_KINDS), ("ms", ctypes.c_double * PROF_KINDS),
                ("alg_bytes", ctypes.c_double * PROF_KINDS), ("s8d_bytes", ctypes.c_double * PROF_KINDS),
                ("flops", ctypes.c_double * PROF_KINDS)]


PtrArr = c_void_p * MAX_TASKS
PtrArr9 = c_void_p * (MAX_TASKS + 1)

_SIGS = {
    "mtlora_version": (c_int, []),
    "mtlora_error_string": (c_char_p, [c_int]),
    "mtlora_roll_and_window_partition_forward": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_void_p]),
    "mtlora_roll_and_window_partition_backward": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_void_p]),
    "mtlora_window_merge_and_roll_forward": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_void_p]),
    "mtlora_window_merge_and_roll_backward": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_void_p]),
    "mtlora_linear_ctx_bytes": (c_int64, [POINTER(LinearDesc)]),
    "mtlora_linear_bwd_scratch_bytes": (c_int64, [POINTER(LinearDesc)]),
    "mtlora_linear_packed_bytes": (c_int64, [POINTER(LinearDesc)]),
    "mtlora_linear_pack": (c_int, [POINTER(LinearDesc), c_void_p, c_void_p, POINTER(c_void_p), POINTER(c_void_p), c_void_p, c_int64,
                                   c_void_p]),
    "mtlora_linear_pack_entry_bytes": (c_int64, []),
    "mtlora_linear_pack_entry": (c_int, [POINTER(LinearDesc), c_void_p, c_void_p, POINTER(c_void_p), POINTER(c_void_p), c_void_p, c_int64,
                                         c_void_p]),
    "mtlora_linear_pack_table": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "mtlora_linear_fwd": (c_int, [POINTER(LinearDesc), c_void_p, POINTER(c_void_p), c_void_p, c_void_p, c_void_p, c_void_p,
                                  POINTER(c_void_p), POINTER(c_void_p), c_void_p, POINTER(c_void_p), c_void_p, c_int64,
                                  c_void_p]),
    "mtlora_linear_bwd": (c_int, [POINTER(LinearDesc), c_void_p, POINTER(c_void_p), c_void_p, c_void_p, POINTER(c_void_p),
                                  c_void_p, c_int64, c_void_p, POINTER(c_void_p), c_void_p, c_void_p, POINTER(c_void_p),
                                  POINTER(c_void_p), c_void_p, c_int64, c_void_p]),
    "mtlora_gemm_tn_scratch_bytes": (c_int64, [c_int64, c_int, c_int]),
    "mtlora_gemm_tn": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int64, c_int64, c_int, c_void_p, c_int64,
                               c_void_p]),
    "mtlora_linear_fwd_gelu": (c_int, [POINTER(LinearDesc), c_void_p, POINTER(c_void_p), c_void_p, c_void_p, c_void_p, c_void_p,
                                       POINTER(c_void_p), POINTER(c_void_p), c_void_p, POINTER(c_void_p), c_void_p,
                                       POINTER(c_void_p), c_void_p, c_int64, c_void_p]),
    "mtlora_linear_bwd_gelu": (c_int, [POINTER(LinearDesc), c_void_p, POINTER(c_void_p), c_void_p, c_void_p, POINTER(c_void_p),
                                       c_void_p, c_int64, c_void_p, POINTER(c_void_p), c_void_p, c_void_p, POINTER(c_void_p),
                                       POINTER(c_void_p), c_void_p, c_int64, c_void_p, POINTER(c_void_p), c_void_p]),
    "mtlora_mlp_hid_supported": (c_int, [POINTER(LinearDesc), POINTER(LinearDesc)]),
    "mtlora_mlp_hid_bwd_scratch_bytes": (c_int64, [POINTER(LinearDesc), POINTER(LinearDesc)]),
    "mtlora_mlp_hid_fwd_scratch_bytes": (c_int64, [POINTER(LinearDesc), POINTER(LinearDesc)]),
    "mtlora_mlp_hid_proj": (c_int, [POINTER(LinearDesc), POINTER(LinearDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "mtlora_mlp_hid_bwd": (c_int, [POINTER(LinearDesc), POINTER(LinearDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, POINTER(c_void_p), POINTER(c_void_p), c_void_p, c_int64, c_void_p]),
    "mtlora_window_attn_bwd_scratch_bytes": (c_int64, [POINTER(AttnDesc)]),
    "mtlora_window_attn_fwd": (c_int, [POINTER(AttnDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mtlora_window_attn_bwd": (c_int, [POINTER(AttnDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_int64, c_void_p]),
    "mtlora_layernorm_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_float,
                                     c_int, c_int, c_int, c_int, c_void_p]),
    "mtlora_layernorm_bwd_scratch_bytes": (c_int64, [c_int64, c_int64, c_int]),
    "mtlora_residual_layernorm_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_void_p, c_int64, c_int64, c_float, c_int, c_int, c_void_p]),
    "mtlora_layernorm_multi_bwd_scratch_bytes": (c_int64, [c_int, c_int64, c_int64, c_int]),
    "mtlora_layernorm_multi_fwd": (c_int, [c_int, POINTER(c_void_p), c_void_p, c_void_p, POINTER(c_void_p), POINTER(c_void_p),
                                           POINTER(c_void_p), c_int64, c_int64, c_float, c_int, c_int, c_int, c_int, c_void_p]),
    "mtlora_layernorm_multi_bwd": (c_int, [c_int, POINTER(c_void_p), POINTER(c_void_p), c_void_p, POINTER(c_void_p),
                                           POINTER(c_void_p), POINTER(c_void_p), c_void_p, c_void_p, c_int64, c_int64, c_int,
                                           c_int, c_void_p, c_int64, POINTER(c_void_p), c_int, c_int, c_void_p]),
    "mtlora_residual_layernorm_streams_fwd": (c_int, [c_int, POINTER(c_void_p), POINTER(c_void_p), c_void_p, c_int64, c_void_p,
                                                      c_void_p, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                                                      POINTER(c_void_p), c_int64, c_int64, c_float, c_int, c_int, c_int, c_int,
                                                      c_void_p]),
    "mtlora_residual_layernorm_streams_bwd": (c_int, [c_int, POINTER(c_void_p), POINTER(c_void_p), c_void_p, POINTER(c_void_p),
                                                      POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), c_void_p, c_void_p,
                                                      c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_void_p, c_int64,
                                                      POINTER(c_void_p), c_int, c_int, c_void_p]),
    "mtlora_residual_layernorm_multi_fwd": (c_int, [c_int, c_void_p, POINTER(c_void_p), c_void_p, c_int64, c_void_p, c_void_p,
                                                    POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                                                    c_int64, c_int64, c_float, c_int, c_int, c_void_p]),
    "mtlora_residual_layernorm_multi_bwd": (c_int, [c_int, POINTER(c_void_p), POINTER(c_void_p), c_void_p, POINTER(c_void_p),
                                                    POINTER(c_void_p), POINTER(c_void_p), c_void_p, POINTER(c_void_p), c_void_p,
                                                    c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_void_p, c_int64,
                                                    c_void_p]),
    "mtlora_residual_layernorm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_void_p, c_int64,
                                              c_void_p, c_void_p]),
    "mtlora_layernorm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_int64, c_int64, c_int, c_int, c_void_p, c_int64, c_void_p, c_int, c_int, c_void_p]),
    "mtlora_bn_scratch_bytes": (c_int64, [c_int64, c_int64, c_int]),
    "mtlora_bn_relu_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_int, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_int64,
                                   c_void_p]),
    "mtlora_bn_relu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                   c_void_p, c_int64, c_int64, c_int, c_void_p, c_int64, c_void_p]),
    "mtlora_residual_droppath_fwd": (c_int, [c_int, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), c_void_p,
                                             c_int64, c_int64, c_int64, c_int, c_int, c_void_p]),
    "mtlora_residual_droppath_bwd": (c_int, [c_int, POINTER(c_void_p), POINTER(c_void_p), c_void_p, c_void_p, c_int64,
                                             c_int64, c_int64, c_int, c_int, c_void_p]),
    "mtlora_upsample_loss_partials": (c_int64, [c_int64, c_int, c_int, c_int]),
    "mtlora_colsum_scratch_bytes": (c_int64, [c_int64, c_int64]),
    "mtlora_colsum": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_int64, c_void_p]),
    "mtlora_label_stat_scratch_bytes": (c_int64, [c_int64]),
    "mtlora_label_stat": (c_int, [c_void_p, c_int64, c_int, c_float, c_void_p, c_void_p, c_int64, c_void_p]),
    "mtlora_upsample_loss": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int,
                                     c_int, c_int, ctypes.c_float, c_void_p]),
    "mtlora_upsample_cl_fwd": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int64, c_int, c_void_p]),
    "mtlora_upsample_cl_bwd": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int64, c_int, c_void_p]),
    "mtlora_block_save_bytes": (c_int64, [POINTER(BlockDesc)]),
    "mtlora_block_fwd_tmp_bytes": (c_int64, [POINTER(BlockDesc)]),
    "mtlora_block_bwd_scratch_bytes": (c_int64, [POINTER(BlockDesc)]),
    "mtlora_block_fwd": (c_int, [POINTER(BlockDesc), POINTER(BlockParams), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                 c_void_p, c_int64, c_void_p]),
    "mtlora_block_bwd": (c_int, [POINTER(BlockDesc), POINTER(BlockParams), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_int64, POINTER(BlockGrads), c_void_p, c_int64, c_int, c_void_p]),
    "mtlora_selftest_layouts": (c_int, [c_void_p, c_void_p]),
    "mtlora_prof_begin": (c_int, [c_int]),
    "mtlora_prof_end": (c_int, [POINTER(ProfSummary)]),
    "mtlora_prof_kind_name": (c_char_p, [c_int]),
}
EXPORTS = tuple(_SIGS)

_lib = None


def lib() -> ctypes.CDLL:
    """Load (once) and return the HIP library; raise loudly if it is missing or stale."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"mtlora_amd: {LIB_PATH} not found. Build it with `python -m mtlora_amd.csrc.build` "
                "(or __graft_entry__.build()). There is no eager/CPU fallback for the MTLoRA hot path.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        v = L.mtlora_version()
        if v != ABI_VERSION:
            raise RuntimeError(f"mtlora_amd: libmtlora_hip.so ABI {v} != binding ABI {ABI_VERSION}; rebuild")
        _lib = L
    return _lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = lib().mtlora_error_string(status).decode()
        raise RuntimeError(f"mtlora_amd: {what} failed: {msg} (status {status})")


def dtype_code(t: torch.Tensor, allow_f16: bool = True) -> int:
    """MTLORA_F32 / BF16 / F16 code of a tensor (the entry points that take no fp16 -- upsample, losses, column sums -- return
    MTLORA_ERR_DTYPE for it; their callers route fp16 tensors around them)."""
    code = _DT.get(t.dtype)
    if code is None or (code == F16 and not allow_f16):
        raise RuntimeError(f"mtlora_amd: unsupported dtype {t.dtype} (fp32, bf16 and fp16 are supported)")
    return code


def require_gpu(*tensors: torch.Tensor) -> None:
    """every tensor on a GPU, and on the CURRENT device: the library launches on ``torch.cuda.current_stream()`` of the
    current device and never calls hipSetDevice, so a tensor living elsewhere would be touched from the wrong device /
    stream (use ``torch.cuda.set_device`` or a ``with torch.cuda.device(x.device)`` block, as for any raw-pointer op)."""
    cur = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("mtlora_amd: tensors must live on a ROCm GPU (MI355X); the HIP path has no CPU fallback")
        if cur is None:
            cur = torch._C._cuda_getDevice()
        if t.device.index != cur:
            raise RuntimeError(f"mtlora_amd: tensor on cuda:{t.device.index} but the current device is cuda:{cur}; the HIP "
                               "path launches on the current device's stream -- call torch.cuda.set_device(tensor.device) first")


def ptr(t) -> c_void_p:
    return c_void_p(0 if t is None else t.data_ptr())


_NO_PTRS = None


def ptr_array(ts) -> "PtrArr":
    global _NO_PTRS
    if not ts:  # (the T = 0 layers pass four empty lists per call: one shared all-null array, never written)
        if _NO_PTRS is None:
            _NO_PTRS = PtrArr()
        return _NO_PTRS
    a = PtrArr()
    for i in range(MAX_TASKS):
        a[i] = 0 if (ts is None or i >= len(ts) or ts[i] is None) else ts[i].data_ptr()
    return a


def ptr_array9(ts) -> "PtrArr9":
    a = PtrArr9()
    for i in range(MAX_TASKS + 1):
        a[i] = 0 if (ts is None or i >= len(ts) or ts[i] is None) else ts[i].data_ptr()
    return a


def stream_ptr() -> c_void_p:
    # (torch.cuda.current_stream() builds a Stream object through four Python layers: ~10 us, once per library call)
    return c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))
