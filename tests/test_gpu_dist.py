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


@pytest.mark.parametrize("world", [2, 3])
def test_score_clips_dp_multi_rank_on_one_card_over_gloo(world):
    """The multi-rank data path with the REAL model on hardware: `world` fresh rank processes on the one MI355X, collectives over gloo
    (RCCL refuses duplicate devices; tests/gloo_two_rank_child.py).  Every rank must reproduce the one-process forward bit for bit -
    which the per-clip / per-frame row plans guarantee for any frame and clip split (round 4)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="4")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "gloo_two_rank_child.py")], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True, env=env, cwd=ROOT))
    outs = [p.communicate(timeout=900) for p in procs]
    for r, (p, (out, err)) in enumerate(zip(procs, outs)):
        print(out[-500:], err[-1500:])
        assert p.returncode == 0, f"rank {r}: {err[-3000:]}"
        assert f"DP_OK rank={r}/{world}" in out


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_batched_eval_loop_on_one_card_over_gloo(world):
    """The reference's eval loop on N ranks as INTEGRATION.md shows it (eval_utils.shard + batched + gather_rows) with the REAL model: `world` rank
    processes on the one MI355X, each scoring its own share in groups of three; every rank ends with the rows of the whole set in the set's order,
    equal to the plain one-clip-per-call loop (tests/gloo_eval_loop_child.py)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="4")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "gloo_eval_loop_child.py")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                      env=env, cwd=ROOT))
    outs = [p.communicate(timeout=900) for p in procs]
    for r, (p, (out, err)) in enumerate(zip(procs, outs)):
        print(out[-300:], err[-1500:])
        assert p.returncode == 0, f"rank {r}: {err[-3000:]}"
        assert f"LOOP_OK rank={r}/{world}" in out


def test_bench_two_ranks_on_one_card_rehearsal():
    """`python bench.py --gpus 2` as a plain command with the REAL model path (tiny dims) and the per-launch roofline pass ON, both ranks
    on the one MI355X over gloo (AIGV_BENCH_SHARE_DEVICE: RCCL refuses duplicate devices): the N > 1 control flow of the bench on hardware -
    self-spawned ranks, settling decisions, the timed region, rank 0's roofline pass with the other ranks in lock-step (round 4 found an
    unmatched barrier there that would have hung every N > 1 run), the per-rank times - ending in ONE json line with n_gpus = 2."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", AIGV_BENCH_SHARE_DEVICE="0", OMP_NUM_THREADS="4")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--model", "tiny", "--clips-per-gpu", "2",
                        "--frames", "2", "--no-cpu-baseline", "--no-decode", "--no-parity", "--motion", "input"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    print(r.stdout[-800:], r.stderr[-1500:])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and (out["process_group"]["backend"], out["process_group"]["world_size"]) == ("gloo", 2) and out["ranks_share_one_device"] is True
    assert len(out["ms_per_step_by_rank"]) == 2 and out["config"]["global_batch_clips"] == 4 and "roofline" in out
