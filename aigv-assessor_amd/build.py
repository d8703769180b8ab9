"""Compile the gfx950 HIP sources into ``libaigv_amd.so`` (in-tree, next to this file).

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting ``.so`` is
git-ignored but travels to the GPU box with the repo snapshot.  Every source is compiled to its own object
(only the stale ones, a few at a time) and the objects are linked: editing one kernel file rebuilds that file.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
SOURCES = ["api.hip", "gemm.hip", "gemm256.hip", "gemmco.hip", "attention.hip", "rowops.hip", "head.hip", "head8.hip", "ingest.hip", "slowfast.hip"]
HEADERS = ["common.h", "kernels.h", "attn_lay.h", os.path.join("..", "..", "include", "aigv_amd.h")]
OUT = os.path.join(HERE, "libaigv_amd.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]
# per-file additions.  The attention kernels' softmax is written one score at a time on purpose (v_pk_*_f32 is slower beside MFMAs): keep
# hipcc's SLP vectoriser from re-packing it
EXTRA_FLAGS = {"attention.hip": ["-fno-slp-vectorize"]}
# diagnostic builds (scripts/*_stamp.py, scripts/gemmco_diag.py): AIGV_HIPCC_DEFINES="-DAIGV_CO_DIAG ..." adds defines to every file; never set for the product
DIAG_DEFINES = os.environ.get("AIGV_HIPCC_DEFINES", "").split()


def _mtime(p: str) -> float:
    return os.path.getmtime(p) if os.path.exists(p) else 0.0


def _obj(src: str) -> str:
    return os.path.join(OBJ, src.replace(".hip", ".o"))


def _stale_objects() -> list:
    newest_header = max(_mtime(os.path.join(CSRC, h)) for h in HEADERS)
    me = _mtime(os.path.abspath(__file__))
    return [s for s in SOURCES if _mtime(_obj(s)) < max(_mtime(os.path.join(CSRC, s)), newest_header, me)]


def _stale() -> bool:
    return bool(_stale_objects()) or _mtime(OUT) < max(_mtime(_obj(s)) for s in SOURCES)


def build(force: bool = False, verbose: bool = False, jobs: int = 4) -> str:
    os.makedirs(OBJ, exist_ok=True)
    todo = list(SOURCES) if force else _stale_objects()
    if not todo and not _stale():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"

    def compile_one(src: str):
        cmd = [hipcc] + FLAGS + DIAG_DEFINES + EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", _obj(src)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        return src, subprocess.run(cmd, capture_output=True, text=True)

    with ThreadPoolExecutor(max_workers=max(1, jobs)) as ex:
        for src, r in ex.map(compile_one, todo):
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed on {src} ({r.returncode}):\n{r.stderr[-4000:]}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + [_obj(s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed ({r.returncode}):\n{r.stderr[-4000:]}")
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
