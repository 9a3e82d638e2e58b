"""Drop-in surface of the part of the reference's ``boundingbox.py`` that is on the hot path."""
import torch

from . import _ffi


def bbox_cxcywh_to_x1y1x2y2(box):
    """In-place cxcywh -> x1y1x2y2 on the last dimension (reference boundingbox.py:25-29), on the GPU.

    Like the reference it writes into ``box`` and returns it.  ``box`` must be a CUDA tensor whose
    last dimension is 4.
    """
    _ffi.require_cuda(box, "box")
    if box.shape[-1] != 4:
        raise _ffi.Yv3Error("last dimension must be 4")
    src = box.float().contiguous()
    n = src.numel() // 4
    with torch.cuda.device(box.device):
        _ffi.check(_ffi.lib().yv3_cxcywh_to_xyxy(src.data_ptr(), src.data_ptr(), n, _ffi.stream_ptr()), "yv3_cxcywh_to_xyxy")
    if src.data_ptr() != box.data_ptr():
        box.copy_(src)
    return box
