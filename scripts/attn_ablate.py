"""Where the pipelined attention kernel (csrc/attention64.hip) spends its time: builds of the library with parts of the kernel
compiled OUT (-DATTN64_ABL=<bits>, see the kernel's header; results are wrong by construction), timed on the two headline shapes.

    python scripts/attn_ablate.py build        # build container: scripts/_abl/libaigv_<bits>.so (hipcc cross-compiles)
    python scripts/attn_ablate.py run          # MI355X: one subprocess per variant -> a table

Compile-time switches, not run-time ones: a run-time branch inside the loop splits its scheduling regions and by itself moved
the kernel from 214 to 249 us.
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "aigv-assessor_amd")
OUT = os.path.join(ROOT, "scripts", "_abl")   # git-ignored; travels to the GPU box with the snapshot (gpurun_out/ does not)
VARIANTS = [(0, "full kernel"), (24, "no MFMA"), (24 + 32, "no MFMA, no LDS fragment reads"), (24 + 64, "no MFMA, no softmax"),
            (24 + 32 + 64 + 128, "no MFMA / LDS reads / softmax / row maxima"), (24 + 32 + 64 + 128 + 4, "... and no DMA in the loop"),
            (24 + 32 + 64 + 128 + 4 + 512 + 1024, "... and no query load, no output store (loop skeleton + barriers)"),
            (256, "one key tile per block (per-block fixed cost + 1 tile)"), (256 + 512 + 1024, "one key tile, no query load, no store"),
            (4, "no DMA in the loop only"), (64, "no softmax only"), (32, "no LDS fragment reads only (MFMAs on stale registers)")]

if sys.argv[1:] == ["build"]:
    sys.path.insert(0, ROOT)
    import importlib
    b = importlib.import_module("aigv_assessor_amd.build")
    b.build()
    os.makedirs(OUT, exist_ok=True)
    objs = [os.path.join(PKG, "build", s.replace(".hip", ".o")) for s in b.SOURCES if s != "attention64.hip"]
    for bits, _ in VARIANTS:
        o = os.path.join(OUT, f"attention64_{bits}.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + b.FLAGS + b.EXTRA_FLAGS.get("attention64.hip", []) + [f"-DATTN64_ABL={bits}", "-c",
                              os.path.join(PKG, "csrc", "attention64.hip"), "-o", o])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUT, f"libaigv_{bits}.so")] + objs + [o])
        os.remove(o)
        print("built", bits)
elif sys.argv[1:] == ["run"]:
    for bits, what in VARIANTS:
        env = dict(os.environ, AIGV_AMD_LIB=os.path.join(OUT, f"libaigv_{bits}.so"), ATTN_ONLY="64")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "attn_bench.py")], env=env, capture_output=True, text=True)
        rows = [ln for ln in r.stdout.splitlines() if "us" in ln and ("32x1024" in ln or "4x2176" in ln)]
        best = {}
        for ln in rows:
            k = ln.split(":")[0].strip()
            us = float(ln.split(":")[1].split("us")[0])
            best[k] = min(best.get(k, 1e9), us)
        print(f"ABL {bits:5d}  " + "  ".join(f"{k}: {v:7.1f} us" for k, v in best.items()) + f"   # {what}", flush=True)
        if r.returncode:
            print(r.stderr[-500:])
