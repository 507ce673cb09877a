"""CPU-side checks of the C ABI: the library loads, exports every symbol include/vortex_hip.h
declares, and its host-side argument checks return the documented codes without touching a GPU."""
import ctypes
import os
import re

import pytest

from vortex_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "vortex_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vx_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(L.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(lib, s), "missing export %s" % s
    assert sorted(L.EXPORTED_SYMBOLS) == syms


def test_version_and_error_strings():
    lib = L.lib()
    assert lib.vx_version() >= 100
    assert lib.vx_error_string(0) == b"ok"
    for code in (-1, -2, -3, -4, -5, -99):
        assert len(lib.vx_error_string(code)) > 0


def test_argument_checks_return_codes_without_gpu():
    lib = L.lib()
    m = L.f3([0, 0, 0])
    # null pointers
    assert lib.vx_preprocess_u8hwc_f32nchw(None, None, 1, 4, 4, 4, 4, m, m, 255, 0, 0, 0, None) == -1
    assert lib.vx_yolo_head_decode(None, None, 1, 13, 80, L.f6([1] * 6), 32.0, 507 * 85, 0, None) == -1
    assert lib.vx_batched_nms(None, None, None, None, None, 1, 10, 6, 0, 0.5, 0, None, None, None, 10, None, None, 0, None) == -1
    # bad dimensions / limits (fake non-null, aligned pointers; nothing is dereferenced on the host)
    p = 1 << 20
    assert lib.vx_preprocess_u8hwc_f32nchw(p, p, 0, 4, 4, 4, 4, m, m, 255, 0, 0, 0, None) == -1
    assert lib.vx_preprocess_u8hwc_f32nchw(p, p, 1, 4, 4, 4, 4, m, m, 255, 0, 7, 0, None) == -1
    assert lib.vx_yolo_head_decode(p, p, 1, 13, 300, L.f6([1] * 6), 32.0, 507 * 305, 0, None) == -2
    assert lib.vx_yolo_head_decode(p, p + 4, 1, 13, 80, L.f6([1] * 6), 32.0, 507 * 85, 0, None) == -5
    assert lib.vx_yolo_head_decode(p, p, 1, 13, 80, L.f6([1] * 6), 32.0, 100, 0, None) == -1
    assert lib.vx_yolo_head_decode3(p, p, None, p, 1, 416, 80, L.f18([1] * 18), None) == -1
    assert lib.vx_yolo_head_decode3(p, p, p, p, 1, 16, 80, L.f18([1] * 18), None) == -1
    assert lib.vx_yolo_head_decode3(p, p, p, p + 4, 1, 416, 80, L.f18([1] * 18), None) == -5
    assert lib.vx_yolo_head_decode3(p, p, p, p, 1, 416, 300, L.f18([1] * 18), None) == -2
    assert lib.vx_yolo_decode(p, 1, 100, 80, 416.0, 0.5, 1, p, p, p, p, p, p, p, 16, None) == -3
    assert lib.vx_batched_nms(p, p, p, p, None, 1, 2_000_000, 6, 0, 0.5, 0, p, p, p, 2_000_000, p, p, 1 << 40, None) == -2
    assert lib.vx_batched_nms(p, p, p, p, None, 1, 10, 6, 0, -0.5, 0, p, p, p, 10, p, p, 1 << 30, None) == -2
    assert lib.vx_batched_nms(p, p, p, p, None, 1, 10, 6, 0, 0.5, 0, p, p, p, 5, p, p, 1 << 30, None) == -1
    assert lib.vx_batched_nms(p, p, p, p, None, 1, 10, 6, 0, 0.5, 0, p, p, p, 10, p, p, 16, None) == -3
    # allocation-free NMS: a call beyond the 64 MiB static scratch is refused on the host
    assert lib.vx_batched_nms_noalloc(p, p, p, p, None, 256, 10647, 6, 80, 0.5, 0, p, p, p, 10647, p, None) == -6
    assert lib.vx_batched_nms_noalloc(p, p, p, p, None, 0, 10, 6, 80, 0.5, 0, p, p, p, 10, p, None) == -1
    assert lib.vx_prior_decode(p, p, 1, 100, 5, 1, 0.1, 0.2, 0.5, 1, p, p, p, p, p, p, p, 1 << 30, None) == -1
    assert lib.vx_prior_decode(p, p, 1, 100, 5, 80, 0.1, 0.2, 0.5, 1, p, p, p, p, p, p, p, 1 << 30, None) == -2


def test_workspace_queries():
    lib = L.lib()
    assert lib.vx_yolo_decode_workspace_bytes(0, 10) == 0
    a = lib.vx_yolo_decode_workspace_bytes(2, 10647)
    assert a >= 2 * 10647 * 32
    assert lib.vx_yolo_fused_workspace_bytes(2, 416) >= 2 * 10647 * 32
    assert lib.vx_batched_nms_workspace_bytes(1, 2_000_000) == 0
    assert lib.vx_batched_nms_workspace_bytes(2, 10647) >= 2 * 16384 * 36
    # round 3: no staged records any more, only the ticket counters (one 256-byte line per image) and one hand-off
    # word per group of 512 rows
    w = lib.vx_prior_decode_workspace_bytes(2, 50400, 5)
    assert 2 * (64 + (50400 + 511) // 512) * 4 <= w < 4096


def test_product_modules_refuse_cpu_tensors():
    """No CPU fallback: the product path fails loudly off-GPU."""
    import torch
    import vortex_amd as V
    with pytest.raises(L.VortexHipError):
        V.Normalize()(torch.zeros(1, 4, 4, 3, dtype=torch.uint8))
    with pytest.raises(L.VortexHipError):
        V.YoloV3PostProcess(img_size=64)(torch.zeros(1, 252, 85), torch.tensor([0.5]), torch.tensor([0.5]))
    with pytest.raises(L.VortexHipError):
        V.batched_nms(torch.zeros(3, 4), torch.zeros(3), torch.zeros(3, dtype=torch.int64), 0.5)


def test_reference_error_behaviour_on_host_checks():
    import torch
    import vortex_amd as V
    with pytest.raises(RuntimeError):          # postprocess/yolov3.py:37-39
        V.YoloV3Decoder(416)(torch.zeros(10, 85), torch.tensor([0.5]))
    nms = V.BatchedNMS()
    z = torch.zeros
    with pytest.raises(RuntimeError):          # nms.py:106-108
        nms(detections=z(1, 3, 6), class_indexes=z(1, 3, dtype=torch.int64), bboxes=z(3, 4), scores=z(3), iou_threshold=torch.tensor([0.5]))
    with pytest.raises(RuntimeError):          # nms.py:118-119
        nms(detections=z(2, 3, 6), class_indexes=z(3, dtype=torch.int64), bboxes=z(3, 4), scores=z(3), iou_threshold=torch.tensor([0.5]))
    with pytest.raises(AssertionError):        # normalizer.py:23-26
        V.Normalize()(torch.zeros(4, 4, 3, dtype=torch.uint8))
    pp = V.YoloV3PostProcess(img_size=416)
    assert pp.additional_inputs == (('score_threshold', (1,)), ('iou_threshold', (1,)))
    assert len(pp.forward.__annotations__) and len(V.Normalize().forward.__annotations__)


def test_documented_noalloc_batch_limits_match_the_library():
    """include/vortex_hip.h and INTEGRATION.md quote the largest batch vx_batched_nms_noalloc accepts for a few
    capacities; the workspace layout decides them (no GPU needed: both functions are host arithmetic)."""
    from vortex_amd import _lib as L
    lib = L.lib()
    limit = lib.vx_batched_nms_noalloc_max_bytes()
    assert limit == 64 << 20
    for capacity, b_max in ((10647, 106), (22743, 14), (50400, 7), (100000, 3)):
        assert lib.vx_batched_nms_workspace_bytes(b_max, capacity) <= limit, capacity
        assert lib.vx_batched_nms_workspace_bytes(b_max + 1, capacity) > limit, capacity
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "vortex_hip.h")).read()
    assert "B <= 106" in text and "B <= 7 at 50400" in text
