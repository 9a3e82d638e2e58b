"""yolo_v3_amd -- MI355X-native YOLOv3 inference hot path (drop-in for ydixon/yolo_v3's
``darknet.py`` / ``yololayer.py`` / ``utils.py`` call surface, computed by hand-written HIP
kernels behind the C-ABI in ``include/yv3.h``).

    from yolo_v3_amd import YoloNet, postprocessing, detect
    net = YoloNet((416, 416)).cuda().eval()
    net.loadWeight("yolov3.weights", "darknet")
    det1, det2, det3 = net(imgs.cuda(), None)
    boxes = postprocessing(torch.cat((det1, det2, det3), 1), 80, 0.5, 0.4)
    boxes = detect(net, imgs.cuda())          # same result, fused, one host sync
"""
from . import arch, synth                      # noqa: F401  (no GPU / extension needed)
from ._ffi import Yv3Error, F32, BF16, F32X3, F32H2   # noqa: F401
from .darknet import (YoloNet, Darknet, PreDetectionConvGroup, UpsampleGroup, WeightManager,   # noqa: F401
                      conv_bn_relu, res_layer)
from .yololayer import YoloLayer               # noqa: F401
from .utils import (postprocessing, iou_vectorized, bbox_iou, PostProcessor, letterbox_transforms,    # noqa: F401
                    letterbox_batch, resize_batch, letterbox_image, load_image, clear_postproc_cache)
from .boundingbox import bbox_cxcywh_to_x1y1x2y2, correct_yolo_boxes, letterbox_reverse, rescale_bbox   # noqa: F401
from .detect import detect, Detector, predict  # noqa: F401
from .dist import detect_sharded                # noqa: F401

__version__ = "0.1.0"
