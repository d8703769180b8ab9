"""Randomised run of the frame resize + ingest kernels (aigv_op_frame_resize_ingest) against LIVE Pillow (Image.resize(..., BICUBIC): what the reference's dataset code calls,
internvl/train/dataset.py:702-738) and against the CPU restatement (oracle/resize.py): random input sizes from 8 x 8 to 1200 x 2000 (extreme aspect ratios, up- and down-scaling,
sizes equal to the output), output sizes 448 / 224 / odd ones, 1-3 frames per call, noise / hard edges / constant frames.  Every output byte equal; the fused bf16 NCHW output equals
ToTensor + Normalize + bf16 of Pillow's bytes.  (test infrastructure: uses oracle/.)

    python tests/manual/fuzz_resize.py [n_cases = 80] [seed = 0]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_ops as T  # noqa: E402
from aigv_assessor_amd import native  # noqa: E402
from oracle import oracle as O  # noqa: E402
from oracle.resize import resize_bicubic_u8  # noqa: E402

try:
    from PIL import Image
except Exception:
    Image = None
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 80
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lib = native.load()
rng = np.random.default_rng(seed0)
bad = 0
print("Pillow:", None if Image is None else Image.__version__ if hasattr(Image, "__version__") else "present", flush=True)
for c in range(n_cases):
    kind = rng.integers(0, 4)
    if kind == 0:
        ih, iw = int(rng.integers(8, 1201)), int(rng.integers(8, 2001))
    elif kind == 1:
        ih, iw = int(rng.integers(8, 64)), int(rng.integers(500, 2001))          # a sliver
    elif kind == 2:
        ih, iw = int(rng.integers(500, 1201)), int(rng.integers(8, 64))
    else:
        ih, iw = [(720, 1280), (1080, 1920), (448, 448), (224, 224), (480, 854), (360, 640)][int(rng.integers(0, 6))]
    oh, ow = [(448, 448), (448, 448), (224, 224), (int(rng.integers(8, 500)), int(rng.integers(8, 500)))][int(rng.integers(0, 4))]
    n = int(rng.integers(1, 4))
    fr = rng.integers(0, 256, (n, ih, iw, 3), dtype=np.uint8)
    style = rng.integers(0, 4)
    if style == 1:
        fr[0] = np.where(rng.random((ih, iw, 3)) < 0.5, 0, 255)                   # hard edges: the clipping paths
    elif style == 2:
        fr[-1] = int(rng.integers(0, 256))                                         # a constant frame must stay constant
    try:
        u8, pv = T._resize_ingest(lib, fr, oh, ow)
    except Exception as e:          # sizes the operator refuses are fine as long as it says so
        print(f"case {c}: {ih}x{iw} -> {oh}x{ow} refused: {str(e)[:120]}", flush=True)
        T._KEEP.clear()
        continue
    got = u8.cpu().numpy()
    ok = True
    for i in range(n):
        want = resize_bicubic_u8(fr[i], oh, ow)
        if Image is not None:
            pil = np.asarray(Image.fromarray(fr[i]).resize((ow, oh), Image.BICUBIC))
            if not np.array_equal(pil, want):
                ok = False
                print(f"  frame {i}: the CPU restatement differs from live Pillow in {int((pil != want).sum())} bytes")
            want = pil
        if not np.array_equal(got[i], want):
            ok = False
            print(f"  frame {i}: {int((got[i] != want).sum())} bytes differ (max |d| {int(np.abs(got[i].astype(int) - want.astype(int)).max())})")
        if not torch.equal(pv[i].cpu(), O.normalize_frames_u8(torch.from_numpy(np.ascontiguousarray(want))[None])[0]):
            ok = False
            print(f"  frame {i}: the normalised bf16 output differs")
    T._KEEP.clear()
    if not ok:
        bad += 1
        print(f"CASE {c} FAILED: {n} x {ih}x{iw} -> {oh}x{ow} style {style}", flush=True)
    if c % 20 == 19:
        print(f"case {c + 1}/{n_cases}: failed so far {bad}", flush=True)
assert bad == 0, bad
print(f"FUZZ_RESIZE_OK {n_cases} cases")
