"""Frame resize + ingest throughput: uint8 HWC video frames -> Pillow-exact BICUBIC 448x448 -> normalised bf16
NCHW, against the HBM roofline and against Pillow on one host core.   python scripts/resize_bench.py"""
import ctypes, sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
from aigv_assessor_amd import native
lib = native.load()
mean, std = (ctypes.c_float * 3)(0.485, 0.456, 0.406), (ctypes.c_float * 3)(0.229, 0.224, 0.225)
for (ih, iw, n) in ((720, 1280, 32), (1080, 1920, 32), (2160, 3840, 8), (448, 448, 32)):
    fr = torch.randint(0, 256, (n, ih, iw, 3), dtype=torch.uint8, device='cuda')
    S = 448
    tmp = torch.empty(n * ih * S * 3, dtype=torch.uint8, device='cuda')
    out = torch.empty(n, 3, S, S, dtype=torch.bfloat16, device='cuda')
    call = lambda: native.check(lib.aigv_op_frame_resize_ingest(fr.data_ptr(), n, ih, iw, S, S, mean, std, tmp.data_ptr(), None, out.data_ptr(), None))
    for _ in range(3): call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    # algorithmic bytes: source frames once, the uint8 horizontal result written and read once, bf16 planes written once
    by = n * (ih * iw * 3 + 2 * ih * S * 3 + 3 * S * S * 2)
    line = f"{n} frames {ih}x{iw} -> 448x448: {us:8.1f} us  {us / n:6.2f} us/frame  {by / us / 1e3:7.1f} GB/s algorithmic ({by / 1e6:.0f} MB)"
    try:
        from PIL import Image
        img = Image.fromarray(fr[0].cpu().numpy())
        t = time.perf_counter()
        for _ in range(5): img.resize((S, S))
        line += f"   | Pillow, one host core: {(time.perf_counter() - t) / 5 * 1e3:.2f} ms/frame"
    except ImportError:
        pass
    print(line, flush=True)
