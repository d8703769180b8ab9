"""Process-group bootstrap and the frame/clip data-parallel scorer (one process per GPU, RCCL over xGMI).

``init_dist`` keeps the reference's signature (internvl/dist_utils.py:32-42) but bootstraps
``torch.distributed`` directly — backend 'nccl' IS RCCL on ROCm — instead of going through DeepSpeed
(:51,103).  The reference's eval issues no collectives and would score the full set on every rank
(stage2_eval.py:908-911); the data-parallel scorer below is the MI355X-native replacement:

  1. frames of the whole batch are split evenly over the ranks, independent of clip boundaries
     (a single clip's 8 frames spread over 8 GPUs in latency mode);
  2. every rank runs InternViT + pixel-shuffle on its frames;
  3. ONE all-gather of the pre-projector visual tokens [F_local, 256, 4*Hv] bf16 makes all tokens visible
     everywhere (xGMI is a full mesh: RCCL's all-gather moves each shard once over each peer link); the SlowFast motion
     branch of this rank's clips was started before step 2 on a side stream and runs beside the ViT shard and this collective;
  4. clips are split over the ranks; each rank runs projector + motion token + LLM pass + heads for its clips;
  5. a tiny all-gather returns (score, answer-row argmax) to every rank.
Weights are replicated (8B: 16 GB, 26B: 51 GB << 288 GB HBM): no tensor/pipeline parallelism.
"""
from __future__ import annotations

import os
import subprocess
from datetime import timedelta
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

timeout = timedelta(minutes=60)
# A one-rank group needs no collective.  Set to True to run them anyway (a one-rank RCCL all-gather is a device copy through RCCL's own
# stream and work handle): how the collective path is exercised on a single MI355X (tests/test_gpu_dist.py, bench.py --force-dp).
force_single_rank_collectives = False
# Measurement hook (bench.py's roofline pass only): a list -> every device all-gather issued through all_gather_rows_begin is completed AT
# ONCE (no overlap with compute in that pass) between two events on the compute stream and booked here as (receive-buffer bytes, start, end);
# read_collective_timing() turns the list into milliseconds.  None (the default, and always inside a timed region): nothing is recorded.
collective_timing = None


def read_collective_timing():
    """[(bytes of the gathered buffer, milliseconds)] of the all-gathers booked since ``collective_timing`` was set to a list; clears it."""
    global collective_timing
    if not collective_timing:
        return []
    torch.cuda.synchronize()
    out = [(n, a.elapsed_time(b)) for n, a, b in collective_timing]
    collective_timing = []
    return out


def init_dist(launcher: str, backend: str = "nccl", **kwargs):
    """internvl/dist_utils.py:32-104: pick the device from the rank and create the default process group."""
    if launcher == "pytorch":
        rank = int(os.environ["RANK"])
        _set_device(int(os.environ.get("LOCAL_RANK", rank)))
    elif launcher == "mpi":
        local_rank = int(os.environ["OMPI_COMM_WORLD_LOCAL_RANK"])
        _set_device(local_rank)
        os.environ.setdefault("MASTER_PORT", "29500")
        if "MASTER_ADDR" not in os.environ:
            raise KeyError("The environment variable MASTER_ADDR is not set")
        os.environ["WORLD_SIZE"] = os.environ["OMPI_COMM_WORLD_SIZE"]
        os.environ["RANK"] = os.environ["OMPI_COMM_WORLD_RANK"]
    elif launcher == "slurm":
        proc_id = int(os.environ["SLURM_PROCID"])
        ntasks = int(os.environ["SLURM_NTASKS"])
        _set_device(proc_id)
        if "MASTER_ADDR" not in os.environ:
            node_list = os.environ["SLURM_NODELIST"]
            os.environ["MASTER_ADDR"] = subprocess.getoutput(f"scontrol show hostname {node_list} | head -n1")
        port = kwargs.pop("port", None)
        if port is not None:
            os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ["WORLD_SIZE"] = str(ntasks)
        os.environ["RANK"] = str(proc_id)
    else:
        raise ValueError(f"Invalid launcher type: {launcher}")
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, timeout=kwargs.pop("timeout", timeout), **kwargs)


def _set_device(idx: int):
    n = torch.cuda.device_count()      # does not initialise the GPU
    if n > 0:
        torch.cuda.set_device(idx % n)


def even_split(n: int, world: int) -> List[Tuple[int, int]]:
    """[lo, hi) of n items for each of `world` ranks; the first n % world ranks take one extra item."""
    q, r = divmod(n, world)
    out, lo = [], 0
    for i in range(world):
        hi = lo + q + (1 if i < r else 0)
        out.append((lo, hi))
        lo = hi
    return out


class GatheredRows:
    """Result of an all-gather along dim 0 with per-rank row counts: ONE buffer [world * max(counts), ...] filled by a single
    ``all_gather_into_tensor`` (rank k's rows at block k, padding behind them when the counts are ragged).  ``rows(lo, hi)`` returns global
    rows [lo, hi) - a VIEW of the buffer when they lie inside one rank's block (always, when the counts are equal), else the
    concatenation of the pieces: only the rows a consumer asks for are ever copied."""

    def __init__(self, buf: torch.Tensor, counts: List[int]):
        self.buf, self.counts, self.m = buf, list(counts), max(counts) if counts else 0
        self.starts = [0]
        for c in counts:
            self.starts.append(self.starts[-1] + c)

    def __len__(self):
        return self.starts[-1]

    def rows(self, lo: int, hi: int) -> torch.Tensor:
        if len(set(self.counts)) == 1:
            return self.buf[lo:hi]
        pieces = []
        for k, c in enumerate(self.counts):
            a, b = max(lo, self.starts[k]), min(hi, self.starts[k] + c)
            if a < b:
                pieces.append(self.buf[k * self.m + a - self.starts[k]: k * self.m + b - self.starts[k]])
        if not pieces:
            return self.buf[:0]
        return pieces[0] if len(pieces) == 1 else torch.cat(pieces, dim=0)

    def all(self) -> torch.Tensor:
        return self.rows(0, len(self))


def all_gather_rows_begin(local: torch.Tensor, counts: List[int], group=None):
    """Start an all-gather along dim 0 with per-rank row counts and return a function that completes it and hands back a
    ``GatheredRows``.  Equal or ragged counts, it is ONE ``all_gather_into_tensor`` on a (padded) buffer - no tensor lists, no
    re-assembly pass.  Between the two calls the collective runs on RCCL's own stream: work enqueued on the compute stream in the
    meantime overlaps with it (the completion only makes the compute stream wait, the host does not block)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and not (force_single_rank_collectives and dist.is_initialized()):
        return lambda: GatheredRows(local, [local.shape[0]])
    tail = tuple(local.shape[1:])
    m = max(counts)
    send = local.contiguous()
    if send.shape[0] != m:                      # ragged: this rank's block is padded to the longest
        pad = torch.zeros((m,) + tail, dtype=local.dtype, device=local.device)
        pad[: send.shape[0]] = send
        send = pad
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # REHEARSAL transport (tests/test_gpu_dist.py: several ranks on ONE card, where RCCL refuses duplicate devices): gloo has no
        # device all-gather, so the shard goes through host memory.  Same partitioning, same result; never used with backend 'nccl'.
        out_h = torch.empty((world * m,) + tail, dtype=local.dtype)
        dist.all_gather_into_tensor(out_h, send.cpu(), group=group)
        out = out_h.to(local.device)
        return lambda: GatheredRows(out, counts)
    out = torch.empty((world * m,) + tail, dtype=local.dtype, device=local.device)
    if collective_timing is not None and local.is_cuda:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dist.all_gather_into_tensor(out, send, group=group, async_op=True).wait()
        e1.record()
        collective_timing.append((out.numel() * out.element_size(), e0, e1))
        return lambda: GatheredRows(out, counts)
    work = dist.all_gather_into_tensor(out, send, group=group, async_op=True)

    def finish():
        work.wait()
        return GatheredRows(out, counts)
    return finish


def all_gather_rows(local: torch.Tensor, counts: List[int], group=None) -> torch.Tensor:
    return all_gather_rows_begin(local, counts, group)().all()


def score_clips_dp(model, pixel_values: torch.Tensor, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor],
                   image_flags: Optional[torch.Tensor], labels: torch.Tensor, motion_feature: Optional[torch.Tensor],
                   mos: Optional[torch.Tensor] = None, group=None, prefer_gathered: bool = False) -> Dict[str, torch.Tensor]:
    """Score a batch of B clips (F frames in total) over all ranks of `group`; every rank passes the same host
    tensors and gets the full result: {'score1' [B], 'logit' [B*(N-1)], 'label' [B*(N-1)]}.

    `model` is an InternVLChatModel (or any object with vit_tokens / forward(visual_tokens=...) / device / stage).
    ``prefer_gathered``: always feed the projector / LLM pass from the ALL-GATHERED token buffer, also where the rank's own shard would
    do (it then waits for the collective instead of overlapping it) - the form in which the collective's output is what gets scored."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B, N = input_ids.shape
    F_total = pixel_values.shape[0]
    if F_total % B:
        raise ValueError("frames must divide evenly over the clips")
    fpc = F_total // B
    csplit = even_split(B, world)
    clo, chi = csplit[rank]
    sl = slice(clo, chi)
    fl = slice(clo * fpc, chi * fpc)
    # 0. the motion branch of this rank's CLIPS (it needs all frames of a clip) depends on the frames only: start it first - on the
    #    product model it runs on a side stream beside the ViT shard and the all-gather, and is joined where its result is consumed
    fsplit = even_split(F_total, world)
    lo, hi = fsplit[rank]
    motion_l = None
    if hasattr(model, "dp_front") and hi > lo:
        # the product model: SlowFast of this rank's clips beside the ViT of its frame shard, as ONE call - a captured HIP graph when the model
        # replays graphs (InternVLChatModel.enable_graph_replay), else exactly the two eager steps below
        need_motion = chi > clo and motion_feature is None
        local, motion_l = model.dp_front(pixel_values[lo:hi], pixel_values[fl] if need_motion else None, chi - clo)
        if chi > clo and motion_feature is not None:
            motion_l = motion_feature[sl]
    else:
        if chi > clo:
            motion_l = motion_feature[sl] if motion_feature is not None else \
                getattr(model, "motion_feature_async", model.motion_feature)(pixel_values[fl], chi - clo)
        # 1-2. frame shard -> ViT tokens
        if hi > lo:
            local = model.vit_tokens(pixel_values[lo:hi])
        else:
            probe = model.vit_tokens(pixel_values[:1])     # keeps shapes/dtypes uniform on idle ranks
            local = probe[:0]
    # 3. all-gather of pre-projector tokens: started here, completed where its result is first needed.  RCCL runs it on its own
    #    stream, so whatever this rank enqueues in between overlaps with it.
    finish_tokens = all_gather_rows_begin(local, [h - l for l, h in fsplit], group)
    # 4. clip shard -> projector + LLM pass.  When the frames of this rank's clips all lie inside its own frame shard (B a multiple
    #    of the world size: the throughput mode of SURVEY.md 8e) the pass needs no remote token and runs while the collective is in
    #    flight; otherwise (a clip's frames spread over ranks: latency mode, ragged splits) it waits for the gathered tokens.
    dev = local.device
    n1 = N - 1
    score_l = torch.zeros((chi - clo,), dtype=torch.float32, device=dev)
    logit_l = torch.full(((chi - clo) * n1,), -1, dtype=torch.long, device=dev)
    own = chi > clo and lo <= clo * fpc and chi * fpc <= hi and not prefer_gathered
    tokens = None if own else finish_tokens()
    if chi > clo:
        vis = local[clo * fpc - lo: chi * fpc - lo] if own else tokens.rows(fl.start, fl.stop)
        out = model(mos=None if mos is None else mos[sl], pixel_values=None, input_ids=input_ids[sl],
                    attention_mask=None if attention_mask is None else attention_mask[sl],
                    image_flags=None if image_flags is None else image_flags[fl], labels=labels[sl],
                    motion_feature=motion_l, visual_tokens=vis)
        logit_l = out["logit"]
        if "score1" in out:
            score_l = out["score1"].float()
    if own:
        finish_tokens()       # every rank ends the step holding every token (the mandated collective), without having waited for it
    # 5. results everywhere
    counts = [h - l for l, h in csplit]
    score = all_gather_rows(score_l, counts, group)
    logit = all_gather_rows(logit_l.view(chi - clo, n1), counts, group).reshape(-1)
    res = {"logit": logit, "label": labels[..., 1:].contiguous().view(-1).to(logit.device)}
    if getattr(model, "stage", 2) == 2:
        res["score1"] = score.to(torch.bfloat16)
    return res
