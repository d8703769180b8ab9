"""world_size-2 test of the frame/clip data-parallel scorer on CPU (gloo): the partitioning, the all-gather of
pre-projector visual tokens and the result gather must reproduce the single-process result exactly.  The model is a
stand-in built from the CPU oracle (the HIP model cannot run here); the same code path drives the real model on RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import aigv_assessor_amd as pkg
from aigv_assessor_amd import dist_utils, synth
from oracle import oracle as O


class OracleBackedModel:
    """Duck-types the three members score_clips_dp uses (vit_tokens / __call__(visual_tokens=...) / stage)."""
    stage = 2

    def __init__(self, cfg, sd, ctx_id):
        self.cfg, self.sd, self.ctx = cfg, sd, ctx_id
        self.device = torch.device("cpu")

    def vit_tokens(self, pv):
        return O.shuffled_tokens(O.vit_forward(self.sd, self.cfg, pv), self.cfg.downsample_ratio)

    def motion_feature(self, pixel_values, clips):
        """Stand-in for the SlowFast branch: a feature that depends on EVERY frame of the clip (so a wrong frame shard shows)."""
        return _clip_feature(pixel_values, clips, self.cfg.motion_dim)

    def __call__(self, mos, pixel_values, input_ids, attention_mask, image_flags, labels, motion_feature, visual_tokens):
        vit = O.projector(self.sd, "mlp1", visual_tokens)
        vit = vit[image_flags.squeeze(-1) == 1]
        mot = O.projector(self.sd, "motion_mlp", motion_feature)
        emb = O.scatter_embeds(self.sd, input_ids, self.ctx, vit, mot)
        hidden, _, _ = O.llm_forward(self.sd, self.cfg, emb, attention_mask)
        logits = O.lm_logits(self.sd, hidden)
        amax = logits[..., :-1, :].argmax(-1)
        keep = labels[:, 1:] != -100
        logit = torch.where(keep, amax, torch.full_like(amax, -1)).reshape(-1)
        return {"logit": logit, "score1": O.score_head(self.sd, self.cfg, hidden[:, -4, :]).squeeze(1)}


def _clip_feature(pixel_values, clips, dim):
    x = pixel_values.float().reshape(clips, -1)
    w = torch.linspace(0.5, 1.5, x.shape[1])
    base = (x * w).mean(dim=1, keepdim=True)
    return (base + torch.linspace(0, 1, dim)[None, :]).to(torch.float32)


def _case(B=3, T=2):
    cfg = pkg.tiny(vit_hidden=64, vit_heads=1, vit_layers=1, vit_inter=128, llm_hidden=256, llm_heads=2, llm_kv_heads=1,
                   llm_layers=1, llm_inter=256, vocab=256, image_size=56, score_dims=(32, 1), motion_dim=128)
    sd = synth.make_state_dict(cfg, seed=5, dtype=torch.float32, rich=True)
    # T = 2, B = 3: 6 frames over 2 ranks = 3+3 (splits clip 1), 3 clips over 2 ranks = 2+1;  B = 1 (latency mode):
    # the clip's frames are split 1+1 and rank 1 has no clip of its own
    toks = synth.canonical_tokens(cfg, B, T, seed=5)
    pv = synth.synthetic_frames(B * T, 56, seed=5, dtype=torch.float32)
    motion = synth.synthetic_motion(B, 128, seed=5, dtype=torch.float32)
    return cfg, sd, toks, pv, motion, B, T


def _worker(rank, world, port, q, use_branch=False, n_clips=3, n_frames=2, prefer_gathered=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    dist_utils.init_dist("pytorch", backend="gloo")
    cfg, sd, toks, pv, motion, B, T = _case(n_clips, n_frames)
    model = OracleBackedModel(cfg, sd, toks["img_context_token_id"])
    out = dist_utils.score_clips_dp(model, pv, toks["input_ids"], toks["attention_mask"], torch.ones(B * T, 1, dtype=torch.long),
                                    toks["labels"], None if use_branch else motion, prefer_gathered=prefer_gathered)
    q.put((rank, out["score1"].float().tolist(), out["logit"].tolist()))
    dist.barrier()
    dist.destroy_process_group()


# world 2: ragged clip / frame splits, latency mode (one clip over two ranks);  world 4: BASELINE config 4's shape (8 clips x 16
# frames: every rank's clips lie inside its own frame shard, so the LLM pass overlaps the token all-gather), a clip count the
# ranks do not divide (5 clips, 10 frames -> 3+3+2+2 frames, 2+1+1+1 clips), and fewer clips than ranks (2 clips on 4 ranks)
@pytest.mark.parametrize("world,use_branch,n_clips,n_frames", [(2, False, 3, 2), (2, True, 3, 2), (2, True, 1, 2), (2, False, 4, 2),
                                                                (4, True, 8, 16), (4, False, 5, 2), (4, True, 2, 4), (2, False, -4, 2),
                                                                (3, True, 1, 8), (3, False, -1, 8)])     # one 8-frame clip over three ranks: 3 + 3 + 2 frames
def test_frame_dp_equals_single_process(world, use_branch, n_clips, n_frames):
    """use_branch: motion_feature=None - every rank runs the model's own motion branch on the frames of ITS clips (the native SlowFast
    branch in the product; a frame-dependent stand-in here)."""
    prefer_gathered = n_clips < 0        # (a negative clip count marks the case that scores from the all-gathered buffer on every rank)
    n_clips = abs(n_clips)
    cfg, sd, toks, pv, motion, B, T = _case(n_clips, n_frames)
    if use_branch:
        motion = _clip_feature(pv, B, cfg.motion_dim)
    ref = O.forward_eval(sd, cfg, pv, toks["input_ids"], toks["attention_mask"], torch.ones(B * T, 1, dtype=torch.long),
                         toks["labels"], motion, toks["img_context_token_id"], stage=2)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, use_branch, n_clips, n_frames, prefer_gathered)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want_logit = torch.where(ref["label"] != -100, ref["logit"], torch.full_like(ref["logit"], -1)).tolist()
    for rank, score, logit in res:
        assert logit == want_logit, f"rank {rank}"
        assert torch.allclose(torch.tensor(score).to(torch.bfloat16).float(), ref["score1"].to(torch.bfloat16).float(), atol=1e-6), (score, ref["score1"])


def test_even_split_and_init_dist_errors():
    assert dist_utils.even_split(8, 8) == [(i, i + 1) for i in range(8)]
    assert dist_utils.even_split(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert sum(h - l for l, h in dist_utils.even_split(7, 3)) == 7
    with pytest.raises(ValueError):
        dist_utils.init_dist("bogus")


def test_bench_multi_process_control_flow_dry_run():
    """bench.py's N > 1 path (process group, barriers around the timed region, all_reduce(MAX) of the rank times, the lock-step
    second pass, score_clips_dp's collectives) executed end to end with torch.distributed.run on gloo / CPU and a stand-in model
    (`--dry-run-cpu`): the launch line is the driver's, with 2 ranks.  A rehearsal of the control flow, not a measurement."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run-cpu",
           "--clips-per-gpu", "2", "--frames", "2"]
    env = dict(os.environ, OMP_NUM_THREADS="2")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]            # rank 0 prints ONE json line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["dry_run"] is True
    assert out["config"]["global_batch_clips"] == 4 and out["scaling"] == "weak" and out["value"] > 0


def test_bench_spawns_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2 --dry-run-cpu` as a PLAIN command (no torch.distributed.run, no RANK / WORLD_SIZE in the environment;
    VERDICT r3 item 2): the process starts its two rank processes itself and rank 0 prints ONE json line with n_gpus 2, the time of every
    rank and the process group's own world size.  A failing rank makes the whole command fail."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run-cpu", "--clips-per-gpu", "2",
           "--frames", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry_run"] is True and out["config"]["global_batch_clips"] == 4
    assert len(out["ms_per_step_by_rank"]) == 2 and all(t > 0 for t in out["ms_per_step_by_rank"])
    pg = out["process_group"]
    assert (pg["backend"], pg["world_size"], pg["distinct_devices"]) == ("gloo", 2, 2)
    assert [d["rank"] for d in pg["devices"]] == [0, 1] and len({d["device"] for d in pg["devices"]}) == 2     # every rank's own device, gathered through the group
    assert abs(out["ms_per_step"] - max(out["ms_per_step_by_rank"])) < 1e-6          # the headline time is the MAX over the ranks
    # a rank that dies takes the command down with a non-zero exit code (an option the ranks reject)
    bad = subprocess.run(cmd + ["--precision", "nonsense"], capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert bad.returncode != 0
    # ... also when the surviving rank ignores SIGTERM (a rank inside a collective whose peer died): SIGKILL after ten seconds
    import time
    t0 = time.time()
    stuck = subprocess.run(cmd + ["--dry-run-fault", "stuck-peer"], capture_output=True, text=True, timeout=200, env=env, cwd=root)
    assert stuck.returncode == 3 and time.time() - t0 < 120, (stuck.returncode, time.time() - t0, stuck.stderr[-500:])


def _gather_worker(rank, world, port, q, counts):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist_utils.init_dist("pytorch", backend="gloo")
    lo = sum(counts[:rank])
    local = (torch.arange(lo, lo + counts[rank], dtype=torch.float32)[:, None] * 10 + torch.arange(3, dtype=torch.float32)[None, :])
    g = dist_utils.all_gather_rows_begin(local, counts)()
    n = sum(counts)
    whole = g.all()
    inside = g.rows(lo, lo + counts[rank]) if counts[rank] else None
    q.put((rank, whole.tolist(), g.buf.shape[0], len(g), g.rows(1, n - 1).tolist(),
           None if inside is None else inside.untyped_storage().data_ptr() == g.buf.untyped_storage().data_ptr()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("counts", [[3, 3], [3, 2], [2, 0], [4, 4, 3, 3]])
def test_ragged_all_gather_is_one_padded_collective(counts):
    """all_gather_rows_begin: equal or ragged per-rank row counts through ONE all_gather_into_tensor on a padded buffer;
    rows inside one rank's block come back as views of that buffer."""
    world = len(counts)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, q, counts)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n = sum(counts)
    want = (torch.arange(n, dtype=torch.float32)[:, None] * 10 + torch.arange(3, dtype=torch.float32)[None, :])
    for rank, whole, buf_rows, length, mid, is_view in res:
        assert whole == want.tolist() and length == n and buf_rows == world * max(counts)
        assert mid == want[1:n - 1].tolist()
        assert is_view in (None, True)
