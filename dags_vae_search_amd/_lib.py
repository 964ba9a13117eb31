"""ctypes binding of libdvs_hip.so (C ABI: include/dvs.h).

The product has exactly one compute path: the HIP library.  ``load()`` raises if it is missing — there is
no CPU fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import (POINTER, Structure, c_char, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_uint32,
                    c_uint64, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libdvs_hip.so"

D_MODEL, HEADS, LAYERS, LATENT, FC_HIDDEN, EMB = 64, 8, 3, 32, 32, 32
MAX_TOKENS = 48           # 16 on the one-tile path (a wavefront owns a DAG); up to 48 on the tiled wide path
TILE_TOKENS = 16
DECODE_STATE_BYTES = 440
CLIP_SCRATCH_FLOATS = 4096  # DVS_CLIP_SCRATCH_FLOATS
RECORD_BYTES = 96         # one-tile path; record_bytes(lib, shape) gives the size that applies
LOSS_FLOATS = 5           # DVS_LOSS_FLOATS: total, recon, kld, non-finite flag, invalid-features flag
ABI_VERSION = 202         # DVS_VERSION of include/dvs.h this binding was written against


class DvsShape(Structure):
    _fields_ = [("batch", c_int32), ("n_tokens", c_int32), ("n_classes", c_int32), ("training", c_int32),
                ("dropout", c_float), ("beta", c_float), ("eps_scale", c_float), ("dag_offset", c_uint32),
                ("seed", c_uint64)]


class DvsParamEntry(Structure):
    _fields_ = [("name", c_char * 64), ("offset", c_int64), ("rows", c_int32), ("cols", c_int32)]


def bind(lib: ctypes.CDLL) -> ctypes.CDLL:
    """Declare argument/return types of every entry point of include/dvs.h on a loaded library."""
    P = POINTER
    lib.dvs_version.restype = c_int
    lib.dvs_last_error.restype = c_char_p
    lib.dvs_device_cus.restype = c_int
    lib.dvs_param_count.restype = c_int64
    lib.dvs_param_count.argtypes = [P(DvsShape)]
    lib.dvs_param_table.restype = c_int
    lib.dvs_param_table.argtypes = [P(DvsShape), P(DvsParamEntry), c_int]
    lib.dvs_workspace_bytes.restype = c_size_t
    lib.dvs_workspace_bytes.argtypes = [P(DvsShape)]
    lib.dvs_record_bytes.restype = c_size_t
    lib.dvs_record_bytes.argtypes = [P(DvsShape)]
    lib.dvs_pack_features.restype = c_int
    # (shape, label 1-hot, position 1-hot, adjacency, target masks, records, records_bytes, status, stream)
    lib.dvs_pack_features.argtypes = [P(DvsShape), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p,
                                      c_void_p]
    lib.dvs_build_records.restype = c_int
    # (shape, labels, preds, records, records_bytes, status, stream)
    lib.dvs_build_records.argtypes = [P(DvsShape), c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]
    lib.dvs_loss_forward.restype = c_int
    # (shape, records, records_bytes, params, n_params, workspace, workspace_bytes, eps, status, losses, mu, logvar, stream)
    lib.dvs_loss_forward.argtypes = [P(DvsShape), c_void_p, c_size_t, c_void_p, c_int64, c_void_p, c_size_t, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.dvs_loss_forward_notify.restype = c_int
    # (... as dvs_loss_forward ..., logvar, host_tail (pinned, 8 words), host_seq, stream)
    lib.dvs_loss_forward_notify.argtypes = [P(DvsShape), c_void_p, c_size_t, c_void_p, c_int64, c_void_p, c_size_t, c_void_p,
                                            c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_uint32, c_void_p]
    lib.dvs_loss_backward.restype = c_int
    # (shape, records, records_bytes, params, n_params, workspace, workspace_bytes, gcoef, grads, stream)
    lib.dvs_loss_backward.argtypes = [P(DvsShape), c_void_p, c_size_t, c_void_p, c_int64, c_void_p, c_size_t, c_void_p,
                                      c_void_p, c_void_p]
    lib.dvs_loss_backward_sq.restype = c_int
    # (shape, records, records_bytes, params, n_params, workspace, workspace_bytes, gcoef, grads, clip_scratch, stream)
    lib.dvs_loss_backward_sq.argtypes = [P(DvsShape), c_void_p, c_size_t, c_void_p, c_int64, c_void_p, c_size_t, c_void_p,
                                         c_void_p, c_void_p, c_void_p]
    lib.dvs_encode.restype = c_int
    # (shape, records, records_bytes, params, n_params, workspace, workspace_bytes, mu, logvar, stream)
    lib.dvs_encode.argtypes = [P(DvsShape), c_void_p, c_size_t, c_void_p, c_int64, c_void_p, c_size_t, c_void_p, c_void_p,
                               c_void_p]
    lib.dvs_decode.restype = c_int
    # (shape, params, n_params, workspace, workspace_bytes, records, records_bytes, z, uniforms, state, state_bytes, stream)
    lib.dvs_decode.argtypes = [P(DvsShape), c_void_p, c_int64, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p,
                               c_void_p, c_size_t, c_void_p]
    lib.dvs_debug_launch.restype = c_int
    lib.dvs_debug_launch.argtypes = [c_size_t, c_void_p]
    lib.dvs_bic_scores.restype = c_int
    lib.dvs_bic_scores.argtypes = [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p]
    lib.dvs_bic_parent_masks.restype = c_int
    lib.dvs_bic_parent_masks.argtypes = [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.dvs_gp_predict.restype = c_int
    lib.dvs_gp_predict.argtypes = [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, ctypes.c_double, ctypes.c_double,
                                   ctypes.c_double, c_void_p, c_void_p]
    lib.dvs_gp_kernel.restype = c_int
    lib.dvs_gp_kernel.argtypes = [c_int32, c_int32, c_int32, c_void_p, c_void_p, ctypes.c_double, ctypes.c_double, c_void_p,
                                  c_void_p]
    lib.dvs_gp_kernel_backward.restype = c_int
    lib.dvs_gp_kernel_backward.argtypes = [c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, ctypes.c_double,
                                           ctypes.c_double, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.dvs_clip_adam.restype = c_int
    # (n, params, grads, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, max_norm, scratch, guard, stream)
    lib.dvs_clip_adam.argtypes = [c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_float,
                                  c_int64, c_float, c_void_p, c_void_p, c_void_p]
    lib.dvs_clip_adam_from_partials.restype = c_int
    lib.dvs_clip_adam_from_partials.argtypes = lib.dvs_clip_adam.argtypes
    lib.dvs_profile_enable.restype = None
    lib.dvs_profile_enable.argtypes = [c_int]
    lib.dvs_profile_collect.restype = c_int
    lib.dvs_profile_collect.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_int]
    lib.dvs_debug_activation.restype = c_int
    lib.dvs_debug_activation.argtypes = [P(DvsShape), c_void_p, c_int, c_void_p, c_void_p]
    return lib


EXPORTS = ["dvs_version", "dvs_last_error", "dvs_device_cus", "dvs_param_count", "dvs_param_table",
           "dvs_workspace_bytes", "dvs_record_bytes", "dvs_pack_features", "dvs_build_records", "dvs_loss_forward", "dvs_loss_forward_notify", "dvs_loss_backward", "dvs_loss_backward_sq", "dvs_encode", "dvs_decode", "dvs_bic_scores", "dvs_bic_parent_masks", "dvs_gp_predict", "dvs_gp_kernel", "dvs_gp_kernel_backward",
           "dvs_clip_adam", "dvs_clip_adam_from_partials", "dvs_debug_activation", "dvs_debug_launch", "dvs_profile_enable", "dvs_profile_collect"]


def profile_collect(lib):
    """{kernel name: (launches, total ms)} recorded since dvs_profile_enable(1)."""
    cap, stride = 64, 64
    names = ctypes.create_string_buffer(cap * stride)
    counts = (c_int * cap)()
    ms = (c_float * cap)()
    n = lib.dvs_profile_collect(names, stride, counts, ms, cap)
    out = {}
    for i in range(min(n, cap)):
        out[names.raw[i * stride:(i + 1) * stride].split(b"\0")[0].decode()] = (int(counts[i]), float(ms[i]))
    return out

_lib = None


def lib_path() -> str:
    return os.path.join(_HERE, LIB_NAME)


def load() -> ctypes.CDLL:
    """Load the in-tree HIP library (built by ``__graft_entry__.build()`` / ``make -C csrc``)."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise RuntimeError(
                f"{LIB_NAME} not found at {path}: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"(hipcc --offload-arch=gfx950).  dags_vae_search_amd has no CPU fallback.")
        _lib = bind(ctypes.CDLL(path))
        if _lib.dvs_version() != ABI_VERSION:
            raise RuntimeError(f"{LIB_NAME}: unexpected ABI version {_lib.dvs_version()}")
    return _lib


def check(lib, code: int, what: str):
    if code != 0:
        msg = lib.dvs_last_error()
        raise RuntimeError(f"{what} failed with code {code}: {msg.decode() if msg else ''}")


def make_shape(batch: int, n_tokens: int, n_classes: int, training: bool = False, dropout: float = 0.15,
               beta: float = 0.005, eps_scale: float = 0.01, dag_offset: int = 0, seed: int = 0) -> DvsShape:
    return DvsShape(int(batch), int(n_tokens), int(n_classes), 1 if training else 0, float(dropout), float(beta),
                    float(eps_scale), int(dag_offset) & 0xFFFFFFFF, int(seed) & 0xFFFFFFFFFFFFFFFF)


def record_bytes(lib, shape: DvsShape) -> int:
    n = int(lib.dvs_record_bytes(ctypes.byref(shape)))
    if n == 0:
        check(lib, 1, "dvs_record_bytes")
    return n


def is_wide(n_tokens: int, n_classes: int) -> bool:
    """Shapes beyond one 16-token / 16-class tile take the workgroup-per-DAG kernels (csrc/dvs_wide.h)."""
    return n_tokens > TILE_TOKENS or n_classes > 16


def param_table(lib, shape: DvsShape):
    """[(name, offset, shape tuple)] of the 108 state-dict tensors inside the flat buffer, and its length."""
    entries = (DvsParamEntry * 128)()
    n = lib.dvs_param_table(ctypes.byref(shape), entries, 128)
    if n <= 0:
        check(lib, 1, "dvs_param_table")
    table = []
    for e in entries[:n]:
        shp = (e.rows, e.cols) if e.cols else (e.rows,)
        table.append((e.name.decode(), int(e.offset), shp))
    total = int(lib.dvs_param_count(ctypes.byref(shape)))
    return table, total
