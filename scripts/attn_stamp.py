"""Where the prefill-attention kernel's waves spend their cycles INSIDE the scorer's step: a diagnostic build of the library with
s_memtime stamps around the phases of attn_fwd_kernel (-DAIGV_ATTN_STAMP, csrc/attention.hip), summed over every wave of every launch.

    python scripts/attn_stamp.py build [extra -D flags]   # build container: scripts/_abl/libaigv_stamp.so (hipcc cross-compiles)
    python scripts/attn_stamp.py run                       # MI355X: the 8B scorer, 4 clips x 8 frames, a few steps -> shares per phase
    ATTN_VARIANT_LIB=libaigv_x.so python scripts/attn_stamp.py build -DFLAG   # an un-stamped A/B variant of attention.hip for AIGV_AMD_LIB=... bench.py

The stamps cost ~10 % themselves and drain the LDS queue where they sit: read the SHARES.  The product build never defines the macro.
"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "aigv-assessor_amd")
OUT = os.path.join(ROOT, "scripts", "_abl")
LIB = os.path.join(OUT, os.environ.get("ATTN_VARIANT_LIB", "libaigv_stamp.so"))

if sys.argv[1:2] == ["build"]:
    sys.path.insert(0, ROOT)
    import importlib
    b = importlib.import_module("aigv_assessor_amd.build")
    b.build()
    os.makedirs(OUT, exist_ok=True)
    SRC = os.environ.get("ATTN_VARIANT_SRC", "attention.hip")      # which source the variant flags apply to
    objs = [os.path.join(PKG, "build", s.replace(".hip", ".o")) for s in b.SOURCES if s != SRC]
    o = os.path.join(OUT, "attention_stamp.o")
    flags = sys.argv[2:] if os.environ.get("ATTN_VARIANT_LIB") else ["-DAIGV_ATTN_STAMP"] + sys.argv[2:]     # a named variant library carries only the flags given
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + b.FLAGS + b.EXTRA_FLAGS.get(SRC, []) + flags + ["-c", os.path.join(PKG, "csrc", SRC), "-o", o])
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
    K8 = 0
    dbg = lib.aigv_debug_attn_stamps
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
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 32)()
    assert dbg(buf, 1) == 0
    n_steps = 3
    for _ in range(n_steps):
        step()
    torch.cuda.synchronize()
    assert dbg(buf, 0) == 0
    names = ["waves", "total", "prologue (Q fragments, first DMA issue)", "wait for the tile (vmcnt) + barrier", "DMA issue of the next tile",
             "S^T = K Q^T (fragment reads, MFMAs) + row maximum", "softmax + P V", "epilogue (normalise, store)", "tiles computed"]
    for d, tag in ((0, "d = 64 (InternViT)"), (1, "d = 128 causal (InternLM2)")):
        v = [buf[d * 16 + i] for i in range(9)]
        if not v[0]:
            continue
        print(f"{tag}: {v[0] / n_steps:.0f} waves per step, {v[1] / v[0]:.0f} cycles per wave, {v[8] / v[0]:.1f} tiles per wave, "
              f"{(v[3] + v[4] + v[5] + v[6]) / max(v[8], 1):.0f} cycles per computed tile")
        for i in range(2, 8):
            print(f"   {names[i]:58s} {100.0 * v[i] / v[1]:5.1f} %   ({v[i] / v[0]:8.0f} cycles per wave)")
        print(f"   {'unaccounted (idle waves of ragged blocks, skipped tiles)':58s} {100.0 * (v[1] - sum(v[2:8])) / v[1]:5.1f} %")
