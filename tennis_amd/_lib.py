"""ctypes binding of libtennis_hip.so (include/tennis_hip.h).

The HIP library is the product: there is no CPU fallback.  Importing this
module without the built library, or calling into it without a GPU, raises.
`import torch` happens first so that the library binds to the HIP runtime torch
already loaded (same SONAME libamdhip64.so.7) and device pointers / streams are
interchangeable with torch tensors.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch  # noqa: F401  (must precede CDLL: shares the HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TENNIS_HIP_LIB") or os.path.join(_HERE, "lib", "libtennis_hip.so")   # env: A/B builds while tuning

LAYOUT_NCHW_F32, LAYOUT_NHWC_F16, LAYOUT_NHWC_U8 = 0, 1, 2
ENC_EXACT_WEIGHTS = 1
RNN_GRU, RNN_LSTM = 0, 1
POOL_MAX, POOL_MEAN = 0, 1


class TnParam(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data_host", C.POINTER(C.c_float)), ("numel", C.c_int64)]


class TnKernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("launches", C.c_int), ("ms", C.c_double),
                ("flops", C.c_double), ("bytes", C.c_double)]


_P = C.c_void_p
_SIGS = {
    "tn_version": (C.c_int, []),
    "tn_last_error": (C.c_char_p, []),
    "tn_ctx_create": (C.c_int, [C.c_int, _P, C.c_int, C.POINTER(_P)]),
    "tn_ctx_stream": (_P, [_P]),
    "tn_ctx_sync": (C.c_int, [_P]),
    "tn_ctx_destroy": (C.c_int, [_P]),
    "tn_densenet121_create": (C.c_int, [_P, C.POINTER(TnParam), C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int,
                                        C.POINTER(_P)]),
    "tn_densenet121_create_ex": (C.c_int, [_P, C.POINTER(TnParam), C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.POINTER(_P)]),
    "tn_densenet121_feature_dim": (C.c_int, [_P]),
    "tn_densenet121_workspace_bytes": (C.c_size_t, [_P]),
    "tn_densenet121_forward": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "tn_densenet121_set_pipelined": (C.c_int, [_P, C.c_int]),
    "tn_densenet121_join": (C.c_int, [_P, C.c_int]),
    "tn_densenet121_profile": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.POINTER(TnKernelStat), C.c_int,
                                         C.POINTER(C.c_int)]),
    "tn_densenet121_read_tap": (C.c_int, [_P, C.c_char_p, C.c_int, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "tn_densenet121_destroy": (C.c_int, [_P]),
    "tn_dense_create": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.POINTER(_P)]),
    "tn_dense_forward": (C.c_int, [_P, _P, C.c_int, _P]),
    "tn_dense_destroy": (C.c_int, [_P]),
    "tn_birnn_create": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(TnParam), C.c_int, C.c_char_p, C.c_int,
                                  C.c_int, C.POINTER(_P)]),
    "tn_birnn_forward": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, _P, _P]),
    "tn_birnn_destroy": (C.c_int, [_P]),
    "tn_preproc_create": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "tn_preproc_forward": (C.c_int, [_P, _P, C.c_int, _P]),
    "tn_preproc_destroy": (C.c_int, [_P]),
    "tn_augment_forward": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _P, _P, _P]),
    "tn_to_tensor_normalize": (C.c_int, [_P, _P, C.c_long, C.POINTER(C.c_float), C.POINTER(C.c_float), _P]),
    "tn_jpeg_create": (C.c_int, [_P, C.POINTER(_P)]),
    "tn_jpeg_info": (C.c_int, [_P, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                               C.POINTER(C.c_int)]),
    "tn_jpeg_decode": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_size_t), C.c_int, _P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "tn_jpeg_sync_passes": (C.c_int, [_P]),
    "tn_jpeg_destroy": (C.c_int, [_P]),
    "tn_temporal_pool": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "tn_prf1_update": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P]),
    "tn_head_create": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(TnParam), C.c_int, C.c_char_p, C.c_char_p, C.c_int,
                                 C.c_int, C.POINTER(_P)]),
    "tn_head_forward_backward": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, _P]),
    "tn_head_buffers": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(C.c_int64)]),
    "tn_head_sgd_step": (C.c_int, [_P, C.c_float, C.c_float, C.c_float, C.c_float]),
    "tn_head_read_param": (C.c_int, [_P, C.c_char_p, C.c_int, C.POINTER(C.c_float), C.c_int64, C.POINTER(C.c_int64)]),
    "tn_head_destroy": (C.c_int, [_P]),
    "tn_finetune_create": (C.c_int, [_P, C.POINTER(TnParam), C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.POINTER(_P)]),
    "tn_finetune_forward_backward": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "tn_finetune_buffers": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(C.c_int64)]),
    "tn_finetune_sgd_step": (C.c_int, [_P, C.c_float, C.c_float, C.c_float, C.c_float]),
    "tn_finetune_read_param": (C.c_int, [_P, C.c_char_p, C.c_int, C.POINTER(C.c_float), C.c_int64, C.POINTER(C.c_int64)]),
    "tn_finetune_destroy": (C.c_int, [_P]),
    "tn_gnmt_trainer_create": (C.c_int, [_P, C.POINTER(TnParam), C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "tn_gnmt_trainer_create_ex": (C.c_int, [_P, C.POINTER(TnParam), C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "tn_gnmt_trainer_dropout_mask": (C.c_int, [_P, C.c_int, C.POINTER(_P)]),
    "tn_gnmt_trainer_forward_backward": (C.c_int, [_P, _P, _P, _P, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "tn_gnmt_trainer_buffers": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(C.c_int64)]),
    "tn_gnmt_trainer_set_dropout": (C.c_int, [_P, C.c_float, C.c_uint64]),
    "tn_gnmt_trainer_dropout_masks": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P)]),
    "tn_gnmt_trainer_adam_step": (C.c_int, [_P, C.c_float, C.c_float, C.c_float, C.c_float]),
    "tn_gnmt_trainer_read_param": (C.c_int, [_P, C.c_char_p, C.c_int, C.POINTER(C.c_float), C.c_int64, C.POINTER(C.c_int64)]),
    "tn_gnmt_trainer_destroy": (C.c_int, [_P]),
    "tn_gnmt_create": (C.c_int, [_P, C.POINTER(TnParam), C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "tn_gnmt_create_ex": (C.c_int, [_P, C.POINTER(TnParam), C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "tn_gnmt_encode": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P]),
    "tn_gnmt_beam_search": (C.c_int, [_P, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, _P, _P, _P,
                                      C.POINTER(C.c_int)]),
    "tn_gnmt_decode_seq": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "tn_masked_softmax_ce": (C.c_int, [_P, _P, _P, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P]),
    "tn_gnmt_destroy": (C.c_int, [_P]),
    "tn_dbg_conv1x1": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, _P, C.c_int, _P, C.c_int, C.c_int, C.c_int,
                                 C.c_int, C.c_int, C.c_int]),
    "tn_dbg_conv3x3": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "tn_dbg_pack_conv3x3": (C.c_int, [_P, _P]),
    "tn_dbg_conv1x1_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, _P, C.c_int, _P, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_int, C.c_int]),
    "tn_dbg_conv3x3_dev": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "tn_dbg_dense_layer_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int,
                                         _P, C.c_int]),
    "tn_dbg_linear": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int]),
    "tn_comm_unique_id": (C.c_int, [_P]),
    "tn_comm_create": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, C.POINTER(_P)]),
    "tn_comm_rank": (C.c_int, [_P]),
    "tn_comm_world": (C.c_int, [_P]),
    "tn_comm_device": (C.c_int, [_P]),
    "tn_comm_uses_rccl": (C.c_int, [_P]),
    "tn_allgather_features": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "tn_allreduce_f32": (C.c_int, [_P, _P, C.c_size_t, C.c_int]),
    "tn_allreduce_i64": (C.c_int, [_P, _P, C.c_size_t]),
    "tn_comm_destroy": (C.c_int, [_P]),
    "tn_dbg_pack_strip": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P]),
    "tn_dbg_dense_strip_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "tn_densenet121_input_means": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "tn_npy_writer_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "tn_npy_writer_submit": (C.c_int, [_P, _P, C.c_int, C.c_int, C.POINTER(C.c_char_p), C.c_int]),
    "tn_npy_writer_drain": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "tn_npy_writer_destroy": (C.c_int, [_P]),
    "tn_bn_relu_clamp_fold": (C.c_int, [_P, _P, _P, _P, C.c_int, _P, _P, _P, _P]),
    "tn_round_fp16_calibrated": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_double, _P]),
    "tn_dbg_block7_create": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, C.POINTER(C.c_void_p)]),
    "tn_dbg_block7_run": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "tn_dbg_block7_run_ts": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "tn_dbg_block7_destroy": (None, [_P]),
    "tn_dbg_block14_create": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, C.POINTER(C.c_void_p)]),
    "tn_dbg_block14_run": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "tn_dbg_block14_run_ts": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "tn_dbg_block14_destroy": (None, [_P]),
    "tn_dbg_block28_create": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, C.POINTER(C.c_void_p)]),
    "tn_dbg_block28_run": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "tn_dbg_block28_run_ts": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "tn_dbg_block28_destroy": (None, [_P]),
}

_lib = None


def declared_symbols():
    return sorted(_SIGS)


def load():
    """Load libtennis_hip.so; raise loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `make` (or __graft_entry__.build()). "
                "tennis_amd has no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().tn_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libtennis_hip {what} failed ({rc}): {msg}")


def make_params(params: dict):
    """dict[str, np.ndarray fp32] -> (TnParam array, keepalive list)."""
    keep = []
    arr = (TnParam * len(params))()
    for i, (k, v) in enumerate(params.items()):
        a = np.ascontiguousarray(v, dtype=np.float32)
        name = k.encode()
        keep += [a, name]
        arr[i].name = name
        arr[i].data_host = a.ctypes.data_as(C.POINTER(C.c_float))
        arr[i].numel = a.size
    return arr, keep


class Context:
    """One tn_ctx bound to a torch device and (by default) torch's current stream."""

    def __init__(self, device: int | None = None, stream=None):
        if not torch.cuda.is_available():
            raise RuntimeError("tennis_amd needs a ROCm GPU: torch.cuda.is_available() is False and there is no "
                               "CPU fallback")
        self.lib = load()
        device = default_device() if device is None else int(device)
        self.device = device
        # the process's current device is left alone: the library switches to `device` for the duration of each
        # call (TnDeviceGuard) and torch tensors carry their own device
        if stream is None:
            stream = torch.cuda.current_stream(device)
        self.torch_stream = stream
        h = _P()
        check(self.lib.tn_ctx_create(device, _P(stream.cuda_stream), 0, C.byref(h)), "tn_ctx_create")
        self.handle = h

    def sync(self):
        check(self.lib.tn_ctx_sync(self.handle), "tn_ctx_sync")

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.tn_ctx_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


_default_ctx = {}


def default_device() -> int:
    """The GPU of this process: torch's current device — one process per GPU, so under torchrun that is
    LOCAL_RANK once the launcher (or init_distributed) has called torch.cuda.set_device — else LOCAL_RANK, else 0."""
    if torch.cuda.is_available():
        cur = torch.cuda.current_device()
        if cur != 0 or "LOCAL_RANK" not in os.environ:
            return cur
        lr = int(os.environ["LOCAL_RANK"])
        return lr if lr < torch.cuda.device_count() else cur
    return 0


def default_context(device: int | None = None) -> Context:
    device = default_device() if device is None else int(device)
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


def ptr(t) -> _P:
    return _P(t.data_ptr()) if t is not None else _P(None)
