"""Records Pillow's own BICUBIC resize results as fixtures for the frame-resize path (run in the build container, where PIL
is installed; the fixtures travel, PIL need not):  python tests/golden/make_resize_golden.py

Each case: a seeded uint8 RGB image and `Image.fromarray(img).resize((out_w, out_h))` — the call the reference makes on every
sampled frame (internvl/train/dataset.py:702-738 with max_num = 1, `Image.resize` default = BICUBIC)."""
import os

import numpy as np
import PIL
from PIL import Image

CASES = [  # in_h, in_w, out_h, out_w, kind
    (97, 131, 64, 48, "edges"),       # downscale both axes, hard 0/255 edges (negative lobes -> clip8)
    (60, 80, 112, 112, "noise"),      # upscale both axes
    (150, 200, 56, 56, "noise"),      # 2.7x / 3.6x downscale: 13- and 17-tap rows
    (56, 90, 56, 40, "gradient"),     # height unchanged: horizontal pass only
    (75, 40, 30, 40, "noise"),        # width unchanged: vertical pass only
    (33, 257, 20, 30, "noise"),       # 8.6x downscale on one axis (37 taps)
]


def make(in_h, in_w, kind, rng):
    if kind == "edges":
        return np.where(rng.random((in_h, in_w, 3)) < 0.5, 0, 255).astype(np.uint8)
    if kind == "gradient":
        y, x = np.mgrid[0:in_h, 0:in_w]
        return np.stack([(x * 255 // max(1, in_w - 1)), (y * 255 // max(1, in_h - 1)), ((x + y) % 256)], axis=-1).astype(np.uint8)
    return rng.integers(0, 256, (in_h, in_w, 3), dtype=np.uint8)


def main():
    rng = np.random.default_rng(20260101)
    out = {"pillow_version": np.array(PIL.__version__)}
    for i, (ih, iw, oh, ow, kind) in enumerate(CASES):
        img = make(ih, iw, kind, rng)
        out[f"in{i}"] = img
        out[f"out{i}"] = np.asarray(Image.fromarray(img).resize((ow, oh)))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "resize.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; Pillow", PIL.__version__)


if __name__ == "__main__":
    main()
