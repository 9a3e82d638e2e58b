"""Drop-in surface of the part of the reference's ``boundingbox.py`` that is on the hot path."""
import numpy as np
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


def _unmap(labels, org_w, org_h, new_w, new_h, is_letterbox, xyxy):
    """Shared GPU path of letterbox_reverse / rescale_bbox / correct_yolo_boxes for a [n, >=4] tensor (CPU or
    GPU; the result follows the input device): un-mapped + clipped x1y1x2y2 (xyxy=True) or xywh."""
    if not isinstance(labels, torch.Tensor):
        raise TypeError("Labels must be a numpy array or pytorch tensor")
    was_cpu = not labels.is_cuda
    if was_cpu and not torch.cuda.is_available():
        raise _ffi.Yv3Error("no GPU available: this package has no CPU path")
    src = labels.detach().float().cuda().contiguous() if was_cpu else labels.detach().float().contiguous()
    n = src.shape[0]
    out = torch.empty((n, 4), dtype=torch.float32, device=src.device)
    if n:
        with torch.cuda.device(src.device):
            org = torch.tensor([[int(org_w), int(org_h)]], dtype=torch.int32, device=src.device)
            _ffi.check(_ffi.lib().yv3_correct_boxes(src.data_ptr(), 1, n, src.shape[1], None, org.data_ptr(), int(new_w), int(new_h),
                                                    int(bool(is_letterbox)), int(bool(xyxy)), out.data_ptr(), _ffi.stream_ptr()),
                       "yv3_correct_boxes")
    return out.cpu() if was_cpu else out


def correct_yolo_boxes(bboxes, org_w, org_h, img_w, img_h, is_letterbox=False):
    """x1y1x2y2 boxes in network-input pixels -> xywh in the original image, clipped (reference
    boundingbox.py:139-149).  Returns a new [n,4] tensor; an empty input is returned unchanged."""
    if len(bboxes) == 0:
        return bboxes
    return _unmap(bboxes[..., :4], org_w, org_h, img_w, img_h, is_letterbox, False)


def _unmap_labels(labels, org_w, org_h, new_w, new_h, is_letterbox):
    """torch tensors and numpy arrays alike (the reference accepts both, boundingbox.py:99-104); a copy is returned."""
    if len(labels) == 0:
        return labels
    if isinstance(labels, np.ndarray):
        out = labels.copy()
        out[..., :4] = _unmap(torch.from_numpy(np.ascontiguousarray(labels[..., :4], dtype=np.float32)),
                              org_w, org_h, new_w, new_h, is_letterbox, True).numpy()
        return out
    if not isinstance(labels, torch.Tensor):
        raise TypeError("Labels must be a numpy array or pytorch tensor")
    out = labels.clone()
    out[..., :4] = _unmap(labels[..., :4], org_w, org_h, new_w, new_h, is_letterbox, True)
    return out


def letterbox_reverse(labels, org_w, org_h, new_w, new_h):
    """reference boundingbox.py:95-116: undo the letterbox mapping on columns 0..3 (x1,y1,x2,y2), clip to the image."""
    return _unmap_labels(labels, org_w, org_h, new_w, new_h, True)


def rescale_bbox(labels, org_w, org_h, new_w, new_h):
    """reference boundingbox.py:119-137: undo a plain resize on columns 0..3 (x1,y1,x2,y2), clip to the image."""
    return _unmap_labels(labels, org_w, org_h, new_w, new_h, False)
