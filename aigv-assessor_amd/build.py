"""Compile the gfx950 HIP sources into ``libaigv_amd.so`` (in-tree, next to this file).

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting ``.so`` is
git-ignored but travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["api.hip", "gemm.hip", "gemm256.hip", "attention.hip", "rowops.hip", "head.hip", "ingest.hip", "slowfast.hip"]
HEADERS = ["common.h", "kernels.h", os.path.join("..", "..", "include", "aigv_amd.h")]
OUT = os.path.join(HERE, "libaigv_amd.so")


def _stale() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
           "-o", OUT] + [os.path.join(CSRC, f) for f in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed ({r.returncode}):\n{r.stderr[-4000:]}")
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
