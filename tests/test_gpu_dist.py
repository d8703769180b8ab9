"""RCCL on one MI355X: the N > 1 code path of dist_utils - process-group bootstrap with backend 'nccl' (RCCL),
the async all-gather of visual tokens on device tensors, the result gathers - executed in a FRESH child process with WORLD_SIZE = 1.
The multi-rank semantics (frame / clip partitioning) are covered by the gloo world-2 / world-4 tests of tests/test_dist_cpu.py; an
8-GPU node is only available to the driver's scaling run."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_score_clips_dp_over_rccl_single_rank():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_single_rank_child.py")], capture_output=True, text=True, timeout=900,
                       env=env, cwd=ROOT)
    print(r.stdout[-1500:], r.stderr[-1500:])
    assert r.returncode == 0, r.stderr[-3000:]
    assert "RCCL_OK backend=nccl" in r.stdout
