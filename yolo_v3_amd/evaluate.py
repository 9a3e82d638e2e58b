"""COCO-results writer of the reference's ``evaluate.py``, fed by the GPU hot path.

Mirrors reference evaluate.py:117-121 (``create_results_entry``), :151-195 (``open_json_pred_writer``,
``JsonPredictionWriter``) and :197-206 (``predict_and_process``): same names, arguments, entry layout
(``image_id, category_id, bbox [x,y,w,h] in ORIGINAL-image pixels, score``), same on-disk byte format
(``json.dump(entry, indent=4, separators=(',', ':'))`` joined by ``,`` inside ``[...]``).  Differences:

* the boxes of a whole batch are un-letterboxed / rescaled by ONE ``yv3_correct_boxes`` launch
  (``csrc/prepost.hip``) instead of one torch expression per image;
* ``predict_and_process`` runs the fused GPU ``detect`` (eval mode: conf 0.005, nms 0.45, evaluate.py:201-204);
* a run without a single detection writes ``[]`` (the reference's seek-back-one-byte leaves ``]``, invalid JSON).

``image_id`` comes from the trailing digits of the file name (reference utils.py:294-297); ``category_id`` is the
class index 0..C-1 as in the reference (it never maps to COCO's sparse ids).
"""
import json
import os.path as osp
import re
from collections import OrderedDict
from contextlib import contextmanager

import torch

from . import _ffi


def get_image_id_from_path(image_path):
    """reference utils.py:294-297"""
    image_path = osp.splitext(image_path)[0]
    m = re.search(r'\d+$', image_path)
    return int(m.group())


def create_results_entry(image_id, category_id, bbox, score):
    """reference evaluate.py:117-121"""
    return OrderedDict({"image_id": image_id, "category_id": category_id, "bbox": bbox, "score": score})


class BatchHandler:
    def process_batch(self, sample, predictions):
        raise NotImplementedError


class JsonPredictionWriter(BatchHandler):
    """reference evaluate.py:164-195"""

    def __init__(self, out_path, classes_names, is_letterbox=False):
        self.out_path = out_path
        self.file = open(out_path, 'w')
        self.classes_names = classes_names
        self.is_letterbox = is_letterbox
        self.entries = 0

    def write_start(self):
        self.file.write('[')

    def write_end(self):
        self.file.write(']')
        self.file.close()

    @staticmethod
    def _wh(img):
        # CHW tensors as the reference's ToTensor produces (evaluate.py:182); HWC uint8 arrays also accepted
        s = tuple(img.shape)
        return (s[2], s[1]) if (len(s) == 3 and s[0] in (1, 3) and s[2] not in (1, 3)) else (s[1], s[0])

    def process_batch(self, sample, predictions):
        imgs, org_imgs, img_paths = sample['img'], sample['org_img'], sample['img_path']
        B = len(img_paths)
        preds = [predictions[i] if (i < len(predictions) and predictions[i] is not None) else torch.Tensor() for i in range(B)]
        counts = [int(p.shape[0]) if p.numel() else 0 for p in preds]
        cap = max(counts) if counts else 0
        if cap == 0:
            return
        if not torch.cuda.is_available():
            raise _ffi.Yv3Error("no GPU available: this package has no CPU path")
        img_w, img_h = self._wh(imgs[0])
        # one launch for the whole batch: padded [B, cap, 7] boxes + per-image counts + original sizes
        host = torch.zeros((B, cap, 7), dtype=torch.float32)
        for i, p in enumerate(preds):
            if counts[i]:
                host[i, :counts[i]] = p.detach().float().cpu()[:, :7]
        org = torch.tensor([list(self._wh(o)) for o in org_imgs], dtype=torch.int32)
        dev = torch.device("cuda")
        boxes, cnt, orgd = host.to(dev), torch.tensor(counts, dtype=torch.int32, device=dev), org.to(dev)
        out = torch.empty((B, cap, 4), dtype=torch.float32, device=dev)
        _ffi.check(_ffi.lib().yv3_correct_boxes(boxes.data_ptr(), B, cap, 7, cnt.data_ptr(), orgd.data_ptr(), int(img_w), int(img_h),
                                                int(bool(self.is_letterbox)), 0, out.data_ptr(), _ffi.stream_ptr()),
                   "yv3_correct_boxes")
        xywh = out.cpu()
        for i in range(B):
            image_id = get_image_id_from_path(img_paths[i])
            for j in range(counts[i]):
                res = create_results_entry(image_id, int(host[i, j, 6].item()), xywh[i, j].tolist(), host[i, j, 5].item())
                if self.entries:
                    self.file.write(',')
                json.dump(res, self.file, indent=4, separators=(',', ':'))
                self.entries += 1


@contextmanager
def open_json_pred_writer(out_path, classes_names, is_letterbox=False):
    """reference evaluate.py:151-158"""
    pred_writer = JsonPredictionWriter(out_path, classes_names, is_letterbox)
    try:
        pred_writer.write_start()
        yield pred_writer
    finally:
        pred_writer.write_end()


def predict_and_process(data, net, num_classes, batch_handler=None, obj_conf_thr=0.005, nms_thr=0.45):
    """reference evaluate.py:197-206: eval-mode detection (is_eval=True; the reference's fixed thresholds conf 0.005 /
    nms 0.45 are the defaults) over an iterable of samples ``{'img': [B,3,H,W] tensor, 'org_img': [...],
    'img_path': [...]}``, each batch handed to ``batch_handler``.  The GPU NMS handles up to ~5e5 (row, class)
    candidates per image and fails loudly beyond (untrained / synthetic weights at conf 0.005 can exceed that)."""
    from .detect import detect
    with torch.no_grad():
        for sample in data:
            predictions = detect(net, sample['img'].cuda(), num_classes, obj_conf_thr=obj_conf_thr, nms_thr=nms_thr,
                                 is_eval=True, use_nms=True)
            if predictions == []:
                predictions = [torch.Tensor() for _ in sample['img_path']]
            batch_handler.process_batch(sample, predictions)


def read_image_rgb(path):
    """uint8 RGB [H,W,3] of an image file (the reference reads with ``cv2.imread`` + BGR->RGB, evaluate.py:136-137; cv2 is not a
    dependency here: PIL decodes -- JPEG decoders may differ from OpenCV's by an LSB on some pixels).  ``cv2.imread`` applies the
    file's EXIF orientation by default (IMREAD_COLOR without IMREAD_IGNORE_ORIENTATION); PIL does not, so it is applied here
    (``ImageOps.exif_transpose``): COCO holds JPEGs with an orientation tag, and both the pixels fed to the network and the
    org_w / org_h the boxes are mapped back with must be the rotated image's, as in the reference."""
    import numpy as np
    from PIL import Image, ImageOps
    with Image.open(path) as im:
        im = ImageOps.exif_transpose(im)
        return np.asarray(im.convert("RGB"), dtype=np.uint8).copy()


def generate_results_file(net, target_txt, classes_names, out, bs, dim, is_letterbox=False):
    """reference evaluate.py:208-219: the COCO-results file of a list of images -- ``target_txt`` is the reference's text file with one
    image path per line (a list of paths, or of ``(path, uint8 RGB [H,W,3] array)`` pairs, is accepted too).  Per batch of ``bs``
    images: the evaluation pipeline's input preparation on the GPU (``IaaLetterbox(dim)`` when ``is_letterbox`` else
    ``iaa.Scale(dim)``, then ``ToTensor``: `utils.letterbox_batch(variant="eval" | "scale")`), eval-mode detection (conf 0.005 / nms
    0.45) and the writer.  ``dim`` = (w, h).  Returns the number of result entries written."""
    from .utils import letterbox_batch
    if isinstance(target_txt, str):
        with open(target_txt, 'r') as f:
            items = [line.strip() for line in f.readlines() if line.strip()]
    else:
        items = list(target_txt)
    numclass = len(classes_names)

    def batches():
        for i in range(0, len(items), bs):
            chunk = items[i:i + bs]
            paths = [c if isinstance(c, str) else c[0] for c in chunk]
            imgs = [read_image_rgb(c) if isinstance(c, str) else c[1] for c in chunk]
            x, _ = letterbox_batch(imgs, dim, variant="eval" if is_letterbox else "scale")
            yield {"img": x, "org_img": imgs, "img_path": paths}       # (the writer needs only the ORIGINAL sizes: HWC arrays)

    with open_json_pred_writer(out, classes_names, is_letterbox) as pred_writer:
        predict_and_process(batches(), net, num_classes=numclass, batch_handler=pred_writer)
        return pred_writer.entries
