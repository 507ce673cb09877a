"""ctypes binding of libvortex_hip.so (include/vortex_hip.h).

There is NO fallback: if the library is missing or a call fails, the product path raises.
Torch is used for device memory and streams only (plumbing)."""
import ctypes
import os
from ctypes import c_float, c_int, c_int64, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VORTEX_HIP_LIB") or os.path.join(_HERE, "lib", "libvortex_hip.so")   # override: A/B builds

VX_OK = 0
VX_STATUS_NAN = 1
VX_STATUS_BAD_CLASS = 2
VX_STATUS_RANGE = 4
VX_STATUS_DEFERRED = 8
VX_STATUS_RANGE2 = 16
VX_MAX_CLASSES = 4096
VX_MAX_CANDIDATES = 1048575
VX_RESIZE_STRETCH = 0
VX_RESIZE_LETTERBOX = 1

_F3 = c_float * 3
_F6 = c_float * 6
_F18 = c_float * 18

_SIGNATURES = {
    "vx_version": (c_int, []),
    "vx_error_string": (ctypes.c_char_p, [c_int]),
    "vx_preprocess_u8hwc_f32nchw": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                            _F3, _F3, c_int, c_int, c_int, c_int, c_void_p]),
    "vx_resize_u8hwc": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vx_yolo_head_decode": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, _F6, c_float,
                                    c_int64, c_int64, c_void_p]),
    "vx_yolo_head_decode3": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, _F18, c_void_p]),
    "vx_yolo_decode_workspace_bytes": (c_size_t, [c_int, c_int]),
    "vx_yolo_decode": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_float, c_int,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_size_t, c_void_p]),
    "vx_yolo_fused_workspace_bytes": (c_size_t, [c_int, c_int]),
    "vx_yolo_fused_decode": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, _F18,
                                     c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "vx_yolo_fused_postprocess_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "vx_yolo_fused_postprocess": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, _F18, c_float, c_int,
                                          c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "vx_yolo_pipeline_step": (c_int, [c_void_p, c_void_p, c_int, c_int, _F3, _F3, c_int, c_int, c_int, c_int,
                                      c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, _F18, c_float, c_int,
                                      c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "vx_yolo_fused_postprocess_overlapped": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, _F18, c_float,
                                                     c_int, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                                     c_size_t, c_void_p, c_void_p]),
    "vx_prior_decode_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "vx_prior_decode": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float,
                                c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_size_t, c_void_p]),
    "vx_batched_nms_workspace_bytes": (c_size_t, [c_int, c_int]),
    "vx_batched_nms": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                               c_int, c_float, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                               c_void_p, c_size_t, c_void_p]),
    "vx_detr_decode": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "vx_softmax_max": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "vx_batched_nms_small": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                     c_int, c_float, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                     c_void_p, c_size_t, c_void_p]),
    "vx_batched_nms_noalloc_max_bytes": (c_size_t, []),
    "vx_batched_nms_noalloc": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                       c_int, c_float, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                       c_void_p]),
    "vx_pack_rows": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                             c_int, c_void_p]),
    "vx_format_results": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, ctypes.POINTER(ctypes.c_int32),
                                  ctypes.POINTER(ctypes.c_uint8), c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vx_profile_enable": (c_int, [c_int]),
    "vx_profile_select": (c_int, [ctypes.c_uint]),
    "vx_profile_disable": (c_int, []),
    "vx_profile_slots": (c_int, []),
    "vx_profile_set_iou_counter": (c_int, [c_void_p]),
    "vx_streams_overlap": (c_int, [c_void_p, c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_float),
                                   ctypes.POINTER(c_float)]),
    "vx_batched_nms_stamps_offset": (c_size_t, [c_int, c_int]),
    "vx_batched_nms_bm_counters_offset": (c_size_t, [c_int, c_int]),
    "vx_profile_get": (c_int, [c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(c_int),
                               ctypes.POINTER(c_float), ctypes.POINTER(c_float), ctypes.POINTER(c_float)]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


class VortexHipError(RuntimeError):
    pass


def lib():
    """Load libvortex_hip.so once.  Raises if it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VortexHipError(
                "libvortex_hip.so not found at %s — build it with `make -C vortex_amd/csrc` "
                "(or __graft_entry__.build()); vortex_amd has no CPU fallback" % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(code: int, what: str):
    if code != VX_OK:
        msg = lib().vx_error_string(code).decode()
        raise VortexHipError("%s failed: %s (code %d)" % (what, msg, code))


def stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def require_cuda(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise VortexHipError(
            "%s must live on an MI355X device (got %s); vortex_amd runs HIP kernels only" % (name, t.device))


def f3(v):
    v = [float(x) for x in v]
    if len(v) == 1:
        v = v * 3
    return _F3(*v)


def f6(v):
    return _F6(*[float(x) for x in v])


def f18(v):
    return _F18(*[float(x) for x in v])


def profile_enable(max_launches: int = 4096, only=None):
    """Bracket kernel launches with HIP events; `only` = iterable of kernel-family names to restrict to."""
    check(lib().vx_profile_enable(int(max_launches)), "vx_profile_enable")
    mask = 0xFFFFFFFF
    if only is not None:
        only, mask = set(only), 0
        for i in range(lib().vx_profile_slots()):
            name, n = ctypes.c_char_p(), c_int()
            tot, mn, mx = c_float(), c_float(), c_float()
            check(lib().vx_profile_get(i, ctypes.byref(name), ctypes.byref(n), ctypes.byref(tot), ctypes.byref(mn),
                                       ctypes.byref(mx)), "vx_profile_get")
            if name.value.decode() in only:
                mask |= 1 << i
    check(lib().vx_profile_select(mask), "vx_profile_select")


def profile_disable():
    check(lib().vx_profile_disable(), "vx_profile_disable")


def profile_read():
    """{kernel family: dict(launches, total_ms, avg_ms, min_ms, max_ms)} for families launched since enable."""
    out = {}
    for i in range(lib().vx_profile_slots()):
        name, n = ctypes.c_char_p(), c_int()
        tot, mn, mx = c_float(), c_float(), c_float()
        check(lib().vx_profile_get(i, ctypes.byref(name), ctypes.byref(n), ctypes.byref(tot), ctypes.byref(mn),
                                   ctypes.byref(mx)), "vx_profile_get")
        if n.value:
            out[name.value.decode()] = dict(launches=n.value, total_ms=tot.value, avg_ms=tot.value / n.value,
                                            min_ms=mn.value, max_ms=mx.value)
    return out
