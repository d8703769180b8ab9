"""Where the 256x256 GEMM kernel's waves spend their cycles INSIDE the scorer's step: a diagnostic build of gemm256.hip with s_memtime
stamps (-DAIGV_GEMM_STAMP) around the prologue (tile mapping, first LDS-DMA units, wait for the first of them), the K loop and the epilogue
(incl. the wait for its stores to be acknowledged - what a workgroup's exit waits for), per epilogue kind and K depth class.

    python scripts/gemm_stamp.py build      # build container: scripts/_abl/libaigv_gstamp.so
    python scripts/gemm_stamp.py run        # MI355X: the 8B scorer, 4 clips x 8 frames, 3 steps

The product build never defines the macro."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "aigv-assessor_amd")
OUT = os.path.join(ROOT, "scripts", "_abl")
LIB = os.path.join(OUT, "libaigv_gstamp.so")

if sys.argv[1:2] == ["build"]:
    sys.path.insert(0, ROOT)
    import importlib
    b = importlib.import_module("aigv_assessor_amd.build")
    b.build()
    os.makedirs(OUT, exist_ok=True)
    objs = [os.path.join(PKG, "build", s.replace(".hip", ".o")) for s in b.SOURCES if s != "gemm256.hip"]
    o = os.path.join(OUT, "gemm256_stamp.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + b.FLAGS + ["-DAIGV_GEMM_STAMP"] + sys.argv[2:] + ["-c", os.path.join(PKG, "csrc", "gemm256.hip"), "-o", o])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + [o])
    os.remove(o)
    print("built", LIB)
else:
    os.environ["AIGV_AMD_LIB"] = LIB
    sys.path.insert(0, ROOT)
    import torch
    import aigv_assessor_amd as pkg
    from aigv_assessor_amd import native, synth
    from aigv_assessor_amd.modeling import InternVLChatModel
    lib = native.load()
    dbg = lib.aigv_debug_gemm_stamps
    dbg.restype, dbg.argtypes = ctypes.c_int, [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    cfg = pkg.internvl2_8b()
    B, T = 4, 8
    dev = torch.device("cuda", 0)
    N = synth.canonical_len(cfg, T)
    model = InternVLChatModel(cfg, device=dev, max_clips=B, max_frames=B * T, max_tokens=B * N)
    model.load_state_dict(synth.make_state_dict(cfg, seed=0, device=dev, rich=True))
    toks = synth.canonical_tokens(cfg, B, T, seed=0)
    model.img_context_token_id = toks["img_context_token_id"]
    model.eval()
    pv = synth.synthetic_frames(B * T, cfg.image_size, seed=0).to(dev)
    motion = synth.synthetic_motion(B, cfg.motion_dim, seed=0).to(dev)
    flags = torch.ones(B * T, 1, dtype=torch.long)
    step = lambda: model(mos=None, pixel_values=pv, input_ids=toks["input_ids"], attention_mask=toks["attention_mask"], image_flags=flags,
                         labels=toks["labels"], motion_feature=motion)
    for _ in range(2):
        step()
    buf = (ctypes.c_ulonglong * 128)()
    assert dbg(buf, 1) == 0
    n_steps = 3
    for _ in range(n_steps):
        step()
    assert dbg(buf, 0) == 0
    epi = ["store (wqkv, ViT qkv, mlp1)", "GELU (ViT fc1, mlp1)", "layer-scale + residual (ViT proj, fc2)", "residual (wo, w2)", "SwiGLU (w1|w3)", "patch embed",
           "split-K slices", "?"]
    print("cycles per wave (8 waves per workgroup = one tile); 'epilogue' ends when the wave's stores have been acknowledged")
    for e in range(8):
        for c in range(2):
            v = [buf[(e * 2 + c) * 8 + i] for i in range(8)]
            if not v[0]:
                continue
            w = v[0]
            print(f"{epi[e]:42s} K tiles {'<= 16' if c == 0 else ' > 16'} (mean {v[5] / w:5.1f}): {w / n_steps / 8:8.0f} tiles/step  total {v[1] / w:8.0f}  "
                  f"prologue {v[2] / w:6.0f} ({100 * v[2] / v[1]:4.1f} %)  K loop {v[3] / w:8.0f} ({100 * v[3] / v[1]:4.1f} %, {v[3] / v[5]:6.0f} per K tile)  "
                  f"epilogue {v[4] / w:6.0f} ({100 * v[4] / v[1]:4.1f} %)")
