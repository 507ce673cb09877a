"""Generate golden input/output vectors by running the REFERENCE's own modules.

Run only in the dev container (needs /root/reference); the GPU box and the test
suite only read the .npz files this writes.  Fixtures are data (inputs + expected
outputs); no reference source is copied.

Recipe: the reference package's __init__ chain pulls in
onnx / pytorch_lightning, which are not installed, so empty package stubs with the
real __path__ are registered and the leaf modules imported directly.  torchvision
is absent too: `torchvision.ops.nms` is provided by the oracle's C restatement
(oracle/nms_oracle.c), so everything AROUND the nms call (offset trick and its
dtype, threshold passing, gather, per-image loop, tuple packing) is the reference's
code, while the greedy core stays "parity unpinned".

    python tests/golden/gen_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import vortex_oracle as O  # noqa: E402

R = "/root/reference/src/development/vortex/development"


def _stub(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    m.__package__ = name
    sys.modules[name] = m


_stub("vortex", "/root/reference/src/development/vortex")
_stub("vortex.development", R)
_stub("vortex.development.networks", R + "/networks")
_stub("vortex.development.networks.modules", R + "/networks/modules")
tv = types.ModuleType("torchvision")
tvo = types.ModuleType("torchvision.ops")
tvo.nms = O.nms
tv.ops = tvo
tv.__version__ = "0.10.1"                        # detr.py:10 parses it (pinned torchvision<=0.10.1)
sys.modules["torchvision"] = tv
sys.modules["torchvision.ops"] = tvo
# detr.py:17,20 import easydict and the backbone registry at module scope; DETRPostProcess uses neither.
ed = types.ModuleType("easydict")
ed.EasyDict = dict
sys.modules["easydict"] = ed
bb = types.ModuleType("vortex.development.networks.modules.backbones")
bb.get_backbone = None
sys.modules["vortex.development.networks.modules.backbones"] = bb
_stub("vortex.development.networks.models", R + "/networks/models")
_stub("vortex.development.networks.models.detection", R + "/networks/models/detection")

from vortex.development.networks.modules.preprocess.normalizer import Normalize, FlipNormalize  # noqa: E402
from vortex.development.networks.modules.heads.detection.yolov3 import YoloV3Layer  # noqa: E402
from vortex.development.networks.modules.postprocess.yolov3 import YoloV3Decoder, YoloV3PostProcess  # noqa: E402
from vortex.development.networks.modules.postprocess.utils.nms import batched_nms, BatchedNMS  # noqa: E402
from vortex.development.networks.modules.utils.darknet import yolo_feature_maps  # noqa: E402
from vortex.development.networks.models.detection.detr import DETRPostProcess  # noqa: E402
from vortex.development.networks.modules.postprocess.base_postprocess import SoftmaxPostProcess  # noqa: E402


def np_(t):
    return t.detach().cpu().numpy()


def gen_normalize():
    g = torch.Generator().manual_seed(0)
    img = torch.randint(0, 256, (2, 6, 8, 3), generator=g, dtype=torch.uint8)
    # every u8 value in every channel: pins the whole 256x3 table
    sweep = torch.arange(256, dtype=torch.uint8).view(1, 16, 16, 1).repeat(1, 1, 1, 3)
    out = {"img": np_(img), "sweep": np_(sweep)}
    cfgs = {
        "imagenet": dict(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225]),
        "half": dict(mean=[0.5, 0.5, 0.5], std=[0.5, 0.5, 0.5]),
        "unit": dict(mean=[0.0, 0.0, 0.0], std=[1.0, 1.0, 1.0]),
    }
    for name, kw in cfgs.items():
        out["mean_" + name] = np.asarray(kw["mean"], np.float32)
        out["std_" + name] = np.asarray(kw["std"], np.float32)
        out["norm_img_" + name] = np_(Normalize(**kw)(img).contiguous())
        out["norm_sweep_" + name] = np_(Normalize(**kw)(sweep).contiguous())
        out["flip_img_" + name] = np_(FlipNormalize(**kw)(img).contiguous())
        out["flip_sweep_" + name] = np_(FlipNormalize(**kw)(sweep).contiguous())
    np.savez_compressed(os.path.join(HERE, "normalize.npz"), **out)


def gen_yolo_layer():
    """YoloV3Layer eval on small heads; also the ordering pin of
    tests/models/test_yolo_head.py:82-156 (anchor-major, y, x; stride 32)."""
    out = {}
    g = torch.Generator().manual_seed(1)
    img_size = 416
    fmaps = yolo_feature_maps(img_size)          # [(52,52),(26,26),(13,13)]
    out["fmaps"] = np.asarray(fmaps, np.int64)
    cases = [("g13_c80", fmaps[2], (6, 7, 8), 80, 1), ("g13_c1", fmaps[2], (6, 7, 8), 1, 2),
             ("g26_c3", fmaps[1], (3, 4, 5), 3, 2), ("g52_c2", fmaps[0], (0, 1, 2), 2, 1)]
    for name, grids, mask, C, B in cases:
        layer = YoloV3Layer(img_size=img_size, grids=grids, mask=list(mask),
                            anchors=O.COCO_ANCHORS, n_classes=C).eval()
        x = torch.randn(B, 3 * (5 + C), grids[0], grids[1], generator=g) * 1.5
        with torch.no_grad():
            y = layer(x)
        out["in_" + name] = np_(x)
        out["out_" + name] = np_(y)
        out["mask_" + name] = np.asarray(mask, np.int64)
        out["anchor_wh_" + name] = np_(layer.anchor_wh).reshape(3, 2)
        out["stride_" + name] = np_(layer.stride)
    np.savez_compressed(os.path.join(HERE, "yolo_layer.npz"), **out)


def gen_decoder_and_postprocess():
    """YoloV3Decoder and YoloV3PostProcess (BatchedNMSPostProcess) of the reference on a
    small decoded head: S=64 (G=2,4,8 -> R=252), C=80, B=3."""
    S, C, B = 64, 80, 3
    heads = O.synth_yolo_heads(B, S, C, seed=7)
    pred = O.yolo_head_eval(heads, S)            # decoded head tensor f32[B,252,85]
    out = {"pred": np_(pred), "img_size": np.asarray(S)}
    dec = YoloV3Decoder(img_size=S, threshold=True)
    dec_nothr = YoloV3Decoder(img_size=S, threshold=False)
    for tname, thr in (("t50", 0.5), ("t06", 0.06), ("t00", 0.0), ("t90", 0.9)):
        st = torch.tensor([thr])
        for b in range(B):
            bb, sc, ci, det = dec(pred[b:b + 1], st)
            k = f"dec_{tname}_b{b}_"
            out[k + "bboxes"], out[k + "scores"] = np_(bb), np_(sc)
            out[k + "cls"], out[k + "det"] = np_(ci), np_(det)
    bb, sc, ci, det = dec_nothr(pred[0:1], torch.tensor([0.5]))
    out["dec_nothr_b0_bboxes"], out["dec_nothr_b0_scores"] = np_(bb), np_(sc)
    out["dec_nothr_b0_cls"], out["dec_nothr_b0_det"] = np_(ci), np_(det)

    pp = YoloV3PostProcess(img_size=S, threshold=True, nms=True)
    assert pp.additional_inputs == (("score_threshold", (1,)), ("iou_threshold", (1,)))
    for pname, sthr, ithr in (("p06_20", 0.06, 0.2), ("p00_100", 0.0, 1.0), ("p90_20", 0.9, 0.2),
                              ("p001_45", 0.001, 0.45), ("p00_00", 0.0, 0.0)):
        res = pp(pred, torch.tensor([sthr]), torch.tensor([ithr]))
        assert isinstance(res, tuple) and len(res) == B
        for b in range(B):
            out[f"post_{pname}_b{b}"] = np_(res[b])
        out[f"post_{pname}_thr"] = np.asarray([sthr, ithr], np.float32)
    np.savez_compressed(os.path.join(HERE, "yolo_postprocess.npz"), **out)


def gen_batched_nms():
    """Reference batched_nms / BatchedNMS wrapper (offset trick incl. its fp32 quantisation)
    on random boxes, with landmarks-width detections (D=16) as well as D=6."""
    out = {}
    for name, K, C, D, seed, thr in (("k300_c80", 300, 80, 6, 3, 0.45), ("k500_c3", 500, 3, 16, 4, 0.3),
                                     ("k64_c1", 64, 1, 6, 5, 0.5), ("k1_c1", 1, 1, 6, 6, 0.5)):
        boxes, scores, cls = O.synth_stress_boxes(K, C, seed)
        g = torch.Generator().manual_seed(seed + 100)
        det = torch.randn(1, K, D, generator=g)
        keep = batched_nms(boxes, scores, cls, torch.tensor([thr]))
        res = BatchedNMS()(detections=det, class_indexes=cls, bboxes=boxes, scores=scores,
                           iou_threshold=torch.tensor([thr]))
        out["boxes_" + name], out["scores_" + name], out["cls_" + name] = np_(boxes), np_(scores), np_(cls)
        out["det_" + name], out["keep_" + name], out["res_" + name] = np_(det), np_(keep), np_(res)
        out["thr_" + name] = np.asarray([thr], np.float32)
    # empty input
    keep = batched_nms(torch.zeros(0, 4), torch.zeros(0), torch.zeros(0, dtype=torch.int64), torch.tensor([0.5]))
    out["keep_empty"] = np_(keep)
    np.savez_compressed(os.path.join(HERE, "batched_nms.npz"), **out)


def _reference_test_tensors():
    """pred_bbox / pred_logits literals of the reference's own tests/models/test_detr.py:12-37 (data)."""
    import ast
    tree = ast.parse(open("/root/reference/tests/models/test_detr.py").read())
    vals = {}
    for node in tree.body:
        if isinstance(node, ast.Assign) and getattr(node.targets[0], "id", None) in ("pred_bbox", "pred_logits"):
            vals[node.targets[0].id] = torch.tensor(ast.literal_eval(node.value.args[0]))
    return vals["pred_bbox"], vals["pred_logits"]


def gen_detr():
    """DETRPostProcess of the reference on (a) the tensors of its own test (threshold 0.3,
    tests/models/test_detr.py:124-145) and (b) seeded COCO-shaped head outputs."""
    pp = DETRPostProcess()
    assert pp.additional_inputs == (("score_threshold", (1,)),)
    out = {}
    pb, pl = _reference_test_tensors()
    x = torch.cat((pb, pl), -1)
    out["ref_in"] = np_(x)
    res = pp(x, score_threshold=torch.tensor(0.3))
    for b, r in enumerate(res):
        out[f"ref_t30_b{b}"] = np_(r)
    for name, B, Q, NL, seed in (("coco", 4, 100, 92, 11), ("small", 3, 7, 2, 12), ("wide", 2, 300, 366, 13)):
        x = O.synth_detr_outputs(B, Q, NL, seed)
        out["in_" + name] = np_(x)
        for tname, thr in (("t30", 0.3), ("t70", 0.7), ("t00", 0.0), ("t100", 1.0)):
            res = pp(x, score_threshold=torch.tensor([thr]))
            assert isinstance(res, tuple) and len(res) == B
            for b, r in enumerate(res):
                out[f"{name}_{tname}_b{b}"] = np_(r)
    np.savez_compressed(os.path.join(HERE, "detr.npz"), **out)


def gen_softmax():
    """SoftmaxPostProcess of the reference (classification post-process) on seeded logits, incl. exact ties
    (first index wins), a single class and a 1000-class head."""
    pp = SoftmaxPostProcess()
    out = {}
    g = torch.Generator().manual_seed(21)
    for name, B, NC, scale in (("c10", 5, 10, 3.0), ("c1000", 4, 1000, 2.0), ("c1", 3, 1, 1.0), ("c65", 7, 65, 8.0)):
        x = torch.randn(B, NC, generator=g) * scale
        out["in_" + name], out["out_" + name] = np_(x), np_(pp(x))
    x = torch.randn(6, 130, generator=g)
    x[0, 5] = x[0, 77] = 9.0                      # exact tie across lanes
    x[1, 64] = x[1, 0] = 7.5                      # exact tie inside one lane's stride
    x[2] = 0.25                                   # every class equal
    x[3, 129] = 30.0                              # last class, saturated
    out["in_ties"], out["out_ties"] = np_(x), np_(pp(x))
    np.savez_compressed(os.path.join(HERE, "softmax.npz"), **out)


if __name__ == "__main__":
    torch.manual_seed(0)
    gen_softmax()
    gen_detr()
    gen_normalize()
    gen_yolo_layer()
    gen_decoder_and_postprocess()
    gen_batched_nms()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
