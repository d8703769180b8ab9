"""Cache policy of the GEMM epilogue's stores, A/B inside the scorer's step.

w1|w3 + SwiGLU writes 250 MB and InternViT fc1 + GELU 268 MB per launch through the same 4 MB L2s that hold the K-slabs of the operands; the
next launch reads them from the Infinity Cache either way.  A DIAGNOSTIC build of gemm256.hip (-DAIGV_STORE_POLICY_AB) lets
GemmArgs::variant_sel 5 / 6 / 7 pick `nt` / `sc1` / `sc1 nt` on the staged epilogues' global_store_dwordx4 per launch (schedule = the shipped
one in every arm), so the arms interleave in ONE process on one box at one clock:

    python scripts/store_policy_ab.py build     # build container: scripts/_abl/libaigv_store_ab.so
    python scripts/store_policy_ab.py run       # MI355X: scripts/step_ab.py over gemm256_variant = 0 (plain) / 5 / 6 / 7 on that library

The product build never defines the macro (its store is the plain one)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "aigv-assessor_amd")
OUT = os.path.join(ROOT, "scripts", "_abl")
LIB = os.path.join(OUT, "libaigv_store_ab.so")

if sys.argv[1:2] == ["build"]:
    sys.path.insert(0, ROOT)
    import importlib
    b = importlib.import_module("aigv_assessor_amd.build")
    b.build()
    os.makedirs(OUT, exist_ok=True)
    objs = [os.path.join(PKG, "build", s.replace(".hip", ".o")) for s in b.SOURCES if s != "gemm256.hip"]
    o = os.path.join(OUT, "gemm256_store_ab.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + b.FLAGS + ["-DAIGV_STORE_POLICY_AB"] + sys.argv[2:] + ["-c", os.path.join(PKG, "csrc", "gemm256.hip"), "-o", o])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + [o])
    os.remove(o)
    print("built", LIB)
else:
    env = dict(os.environ, AIGV_AMD_LIB=LIB)
    arms = sys.argv[2:] or ["gemm256_variant=0", "gemm256_variant=5", "gemm256_variant=6", "gemm256_variant=7"]
    raise SystemExit(subprocess.call([sys.executable, os.path.join(ROOT, "scripts", "step_ab.py")] + arms, env=env, cwd=ROOT))
