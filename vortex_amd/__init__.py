"""vortex_amd — MI355X-native detection pre/post-process path of nodefluxio/vortex.

Hand-written HIP kernels (vortex_amd/csrc, C ABI in include/vortex_hip.h) behind the
reference's own nn.Module API for that path:

    preprocess.Normalize / FlipNormalize / get_preprocess
    heads.YoloV3Layer
    postprocess.YoloV3Decoder / BatchedNMS / NMS / NoNMS / BatchedNMSPostProcess /
                YoloV3PostProcess / YoloV3FusedPostProcess / SoftmaxPostProcess / NoOpPostProcess
    prior.PriorBoxDecoder / PriorBoxPostProcess
    predictor.BasePredictor / create_predictor / get_prediction_results
    detr.DETRPostProcess
    runtime.resize_batch / format_detections / adjust_coordinates / InferenceHelper
    parallel.shard_batch / allgather_detections
    YoloV3PipelineStep                   (pre-process + fused post-process in ONE library call: vx_yolo_pipeline_step)
    graphs.GraphedPostProcess            (HIP-graph replay for launch-bound batch sizes; import on demand)
    export                               (torch.library custom ops + ONNX symbolic; import on demand)
"""
from . import _lib, ops, runtime  # noqa: F401
from .preprocess import Normalize, FlipNormalize, get_preprocess  # noqa: F401
from .heads import YoloV3Layer, YoloV3HeadDecode, yolo_feature_maps  # noqa: F401
from .postprocess import (YoloV3Decoder, BatchedNMS, NMS, NoNMS, BasicNMSPostProcess,  # noqa: F401
                          BatchedNMSPostProcess, YoloV3PostProcess, YoloV3FusedPostProcess,
                          NoOpPostProcess, SoftmaxPostProcess, batched_nms, get_postprocess)
from .prior import PriorBoxDecoder, PriorBoxPostProcess  # noqa: F401
from .detr import DETRPostProcess  # noqa: F401
from .predictor import BasePredictor, create_predictor, get_prediction_results  # noqa: F401
from .step import YoloV3PipelineStep  # noqa: F401
from .streams import StreamRing  # noqa: F401

__version__ = "0.1.0"
