"""Drop-in surface of the reference's ``yololayer.py`` (inference branch only).

``YoloLayer.forward(x, img_dim)`` turns head logits ``[B, 3*(5+C), H, W]`` (the reference's NCHW
layout, channel = anchor*(5+C)+attr, yololayer.py:42) into ``[B, H*W*3, 5+C]`` rows
``cx, cy, w, h, conf, cls...`` in input pixels (yololayer.py:45-59,98-104) with ONE fused HIP
kernel (``yv3_decode_nchw``); the reference does the box part on the CPU and crosses the
GPU<->CPU boundary twice (yololayer.py:58-59,98).  Inside ``YoloNet`` the NHWC variant
(``yv3_decode``) is used directly on the head conv's output, with no permute at all.

The training branch (loss / build_target_tensor, yololayer.py:64-95,107-172) is out of scope.
"""
import ctypes

import torch
import torch.nn as nn

from . import _ffi


class YoloLayer(nn.Module):
    def __init__(self, anchors_all, anchors_mask, img_dim, numClass):
        super().__init__()
        self.anchors_all = anchors_all          # list of (w, h) pairs, input pixels
        self.anchors_mask = anchors_mask
        self.img_dim = img_dim
        self.numClass = numClass
        self.bbox_attrib = 5 + numClass
        self.ignore_thres = 0.7                 # kept for attribute parity (training only)

    def forward(self, x, img_dim, target=None):
        if target is not None:
            raise NotImplementedError("training loss (reference yololayer.py:64-95) is outside the inference hot path")
        _ffi.require_cuda(x, "head logits")
        nB, ch, nH, nW = x.shape
        nA = len(self.anchors_mask)
        if nA != 3 or ch != nA * self.bbox_attrib:
            raise _ffi.Yv3Error("expected %d channels (3 anchors x %d), got %d" % (3 * self.bbox_attrib, self.bbox_attrib, ch))
        x = x.float().contiguous()
        stride = img_dim[1] / nH                                            # yololayer.py:36
        flat = []
        for m in self.anchors_mask:
            flat += [float(self.anchors_all[m][0]), float(self.anchors_all[m][1])]
        out = torch.empty((nB, nA * nH * nW, self.bbox_attrib), device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _ffi.check(_ffi.lib().yv3_decode_nchw(x.data_ptr(), (ctypes.c_float * 6)(*flat), stride, out.data_ptr(),
                                                  out.shape[1] * out.shape[2], nB, nH, nW, self.numClass,
                                                  _ffi.stream_ptr()), "yv3_decode_nchw")
        return out
