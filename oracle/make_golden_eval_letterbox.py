"""Golden fixture for the EVALUATION pipeline's input preparation ("eval variant"): geometry of the REFERENCE's
``transforms.IaaLetterbox._compute_height_width_pad`` (transforms.py:196-205) on a table of image shapes, and the value map of its
``ToTensor`` (transforms.py:25-43: ``torch.from_numpy(img).float().permute(2,0,1) / 255.0``) on all 256 byte values.

Build-container only (needs /root/reference).  ``transforms.py`` imports cv2 and imgaug, absent here: both are replaced by empty
stand-in modules (``Augmenter`` becomes an empty base class); what runs for real is the reference's own geometry code and ToTensor.
The RESAMPLING inside ``IaaLetterbox`` is imgaug -> ``cv2.resize(INTER_CUBIC)``: third-party, not runnable here, parity unpinned
(oracle_cpu.cv_resize_cubic_u8 restates OpenCV's fixed-point path).

    python oracle/make_golden_eval_letterbox.py        # rewrites tests/golden/eval_letterbox.npz
"""
import os
import sys
import types
import warnings

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
GOLD = os.path.join(REPO, "tests", "golden")


class _Anything(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return type(name, (), {})


def main():
    sys.path.insert(0, "/root/reference")
    for n in ["cv2", "imgaug", "imgaug.augmenters", "torchvision", "torchvision.transforms", "torchvision.datasets", "torchvision.models"]:
        sys.modules[n] = _Anything(n)
    sys.modules["imgaug"].augmenters = sys.modules["imgaug.augmenters"]
    tv = sys.modules["torchvision"]
    tv.transforms, tv.datasets, tv.models = (sys.modules["torchvision." + k] for k in ("transforms", "datasets", "models"))
    warnings.simplefilter("ignore")
    import torch
    import transforms as ref                                             # the reference's transforms.py

    shapes = [(452, 602), (480, 640), (500, 333), (1080, 1920), (417, 417), (415, 833), (333, 500), (97, 1231), (1231, 97),
              (416, 416), (608, 608), (607, 609), (31, 33), (2000, 3008), (641, 479)]
    dims = [(416, 416), (608, 608), (320, 320), (416, 608), (608, 416)]             # (w, h)
    rows = []
    for (h, w) in shapes:
        for (dw, dh) in dims:
            rw, rh, xp, yp = ref.IaaLetterbox._compute_height_width_pad((h, w, 3), dh, dw)
            rows.append((h, w, dw, dh, rw, rh, xp, yp))
    ramp = np.arange(256, dtype=np.uint8).reshape(16, 16, 1).repeat(3, 2)          # every byte value, 3 channels
    t = ref.ToTensor()({"img": ramp.copy(), "label": None})["img"]
    np.savez_compressed(os.path.join(GOLD, "eval_letterbox.npz"), geometry=np.array(rows, dtype=np.int32), ramp=ramp, ramp_tensor=t.numpy())
    print("wrote eval_letterbox.npz:", len(rows), "geometry rows; ToTensor ramp", tuple(t.shape), t.dtype)


if __name__ == "__main__":
    main()
