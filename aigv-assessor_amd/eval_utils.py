"""Host-side result handling of the eval drivers: answer-token slice, quality-level parse, CSV and
correlation metrics — deterministic string/array code, restated from internvl/train/internvl/eval/stage2_eval.py."""
from __future__ import annotations

import csv
from typing import Optional, Sequence

LEVELS = (("bad", 1), ("poor", 2), ("fair", 3), ("good", 4), ("excellent", 5))
CSV_COLUMNS = ("video_name", "answer", "output", "mos", "pred_score", "level")   # stage2_eval.py:654


def answer_ids(labels_row, logit_row, im_end_id: int = 92542):
    """stage2_eval.py:940-941: answer tokens = labels not in {-100, <|im_end|>}; predictions = logit[-len-1:-1]."""
    n = sum(1 for x in labels_row.tolist() if x != -100 and x != im_end_id)
    return logit_row[-n - 1:-1]


def parse_level(text: str) -> int:
    """stage2_eval.py:956-967: substring tests in the order bad, poor, fair, good, excellent -> 1..5, else 0."""
    for word, level in LEVELS:
        if word in text:
            return level
    return 0


def save_and_evaluate(rows: Sequence[Sequence], output_file: Optional[str] = None) -> dict:
    """stage2_eval.py:652-688: CSV dump, substring accuracy (output in answer), SRCC/PLCC/KRCC of level and of
    pred_score against mos."""
    from scipy.stats import kendalltau, pearsonr, spearmanr
    if output_file:
        with open(output_file, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(CSV_COLUMNS)
            w.writerows(rows)
    total = len(rows)
    acc = sum(1 for r in rows if r[2] in r[1]) / total if total else 0.0
    mos = [float(r[3]) for r in rows]
    out = {"acc": acc}
    for name, col in (("level", 5), ("pred_score", 4)):
        vals = [float(r[col]) for r in rows]
        out[f"{name}_srcc"] = spearmanr(mos, vals)[0]
        out[f"{name}_plcc"] = pearsonr(mos, vals)[0]
        out[f"{name}_krcc"] = kendalltau(mos, vals)[0]
    return out


def lookahead(items, model, frames, n_clips: int = 1):
    """Iterate an eval loop one item ahead: yields ``(item, ahead)`` where ``ahead`` is ``model.prefetch(...)`` of THAT item's frames, started
    while the previous item was still being scored - the next clip's frame ingest, InternViT pass and SlowFast branch run beside the
    current clip's InternLM2 pass.  ``frames(item)`` returns the item's frames: uint8 [F, H, W, 3] (ingested on the GPU) or normalised
    [F, 3, S, S] ``pixel_values``.  The reference's loop (stage2_eval.py:908-941) changes by two lines:

        for item, ahead in eval_utils.lookahead(dataloader, model, frames=lambda it: it["pixel_values"][0]):
            output = model(pixel_values=ahead, input_ids=..., ...)            # instead of pixel_values=item["pixel_values"][0].to(...).cuda()

    Scores and level tokens are those of the plain loop, bit for bit."""
    import torch

    def start(item):
        f = frames(item)
        return model.prefetch(frames_u8=f, n_clips=n_clips) if f.dtype == torch.uint8 else model.prefetch(pixel_values=f, n_clips=n_clips)
    it = iter(items)
    try:
        cur = next(it)
    except StopIteration:
        return
    ahead = start(cur)
    for nxt in it:
        nxt_ahead = start(nxt)          # enqueued BEFORE the caller scores `cur`: the two then run side by side
        yield cur, ahead
        cur, ahead = nxt, nxt_ahead
    yield cur, ahead


# ---- the reference's eval loop at batch k, on N ranks (stage2_eval.py:908-941) -------------------------------------------------------
def shard(items, rank: int, world: int):
    """This rank's share of an eval set: items rank, rank + world, rank + 2 world, ...  The reference launches ``torchrun
    --nproc_per_node=${GPUS}`` (shell/eval/stage2_eval.sh:20-25) with a plain ``DataLoader(train_dataset, batch_size=1)`` and no sampler
    (stage2_eval.py:908-911): with GPUS > 1 every rank scores the WHOLE set.  A map-style dataset (``__len__`` + ``__getitem__``) comes
    back as a ``torch.utils.data.Subset`` - so the other ranks' videos are never decoded here -, any other iterable as a strided
    iterator.  ``gather_rows`` puts the per-rank result rows back into the set's order."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside 0..{world - 1}")
    if world == 1:
        return items
    if hasattr(items, "__len__") and hasattr(items, "__getitem__"):
        from torch.utils.data import Subset
        return Subset(items, range(rank, len(items), world))
    import itertools
    return itertools.islice(items, rank, None, world)


def gather_rows(rows: Sequence, group=None) -> list:
    """The result rows of every rank (each the rows of its ``shard``, in its own order) -> the rows of the whole set in the set's order, on
    every rank.  One ``all_gather_object`` of small host tuples at the END of the loop: nothing crosses ranks while clips are scored."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return list(rows)
    world = dist.get_world_size(group)
    parts = [None] * world
    dist.all_gather_object(parts, list(rows), group=group)
    return _interleave(parts)


def _interleave(parts: Sequence[Sequence]) -> list:
    """The inverse of ``shard``: rank r holds items r, r + world, r + 2 world, ... of the set as its 0th, 1st, 2nd, ... row."""
    out = []
    for i in range(max((len(p) for p in parts), default=0)):
        for p in parts:                                   # item j of the set went to rank j % world as its (j // world)-th
            if i < len(p):
                out.append(p[i])
    return out


def _frames_of(item):
    pv = item["pixel_values"]
    return pv[0] if pv.dim() == 5 else pv                  # DataLoader(batch_size=1) adds the leading 1 (stage2_eval.py:932 takes [0])


def batched(items, model, k: int = 4, frames=None, ahead: bool = True, pad_id: int = 2):
    """The reference's eval loop at batch ``k`` instead of batch 1: yields ``(item, output)`` for EVERY item of ``items`` (the loop's
    ``DataLoader(batch_size=1)`` items: ``input_ids`` / ``attention_mask`` / ``labels`` [1, N_i] with N_i ragged, ``image_flags``
    [1, T, 1], ``pixel_values`` [1, T, 3, S, S] or what ``frames(item)`` returns - uint8 [T, H, W, 3] decoded frames are ingested on the
    GPU), where ``output`` is what ``model(...)`` on that item alone returns - ``score1`` [1], ``logit`` [N_i - 1], ``label`` [N_i - 1], and
    ``loss`` when the item carries ``mos`` - as HOST tensors, bit for bit: a clip's score and level tokens do not depend on its batch
    mates (per-clip row plans; tests/test_gpu_e2e.py proves ``torch.equal``), so grouping changes the speed and nothing else.

    Up to ``k`` consecutive items of the same frame geometry form a group: ids / labels right-padded to the group's longest prompt as
    the reference's collator does (internvl/patch/pad_data_collator.py:60-67: pad id, IGNORE_INDEX, mask = ids != pad - here the items'
    own masks are kept), frames and flags concatenated (:93-99), ONE ``forward`` and ONE device-to-host copy of its results per group.
    With ``ahead`` the NEXT group's visual front (H2D copy, resize + normalise, InternViT, SlowFast) runs on its own stream beside this
    group's InternLM2 pass (``InternVLChatModel.prefetch``), and a group's results are fetched only after the next group's pass has been
    enqueued (the GPU never waits for the host's loop body; results arrive one group late, in order).  The loop changes by two lines:

        for item2, output in eval_utils.batched(eval_utils.shard(train_dataloader, rank, world), model, k=4):
            score1 = output['score1'].item()          # `output = model(...)` and the `.to(model.device)` copies above it go away

    ``items`` may be sharded first (``shard``) so that N ranks score N disjoint shares."""
    import torch
    import torch.nn.functional as F
    if k < 1:
        raise ValueError("k must be >= 1")
    get = frames or _frames_of

    def groups():
        cur, geom = [], None
        for item in items:
            f = get(item)
            g = (tuple(f.shape), f.dtype)
            if cur and (g != geom or len(cur) == k):
                yield cur
                cur = []
            geom = g
            cur.append((item, f))
        if cur:
            yield cur

    def start(group):
        fr = [f for _, f in group]
        if fr[0].dtype == torch.uint8:                     # decoded frames: copied up clip by clip and joined on the device (ingest_frames)
            return model.prefetch(frames_u8=fr, n_clips=len(group)) if ahead else model.ingest_frames(fr)
        pv = torch.cat([f.to(device=model.device, dtype=torch.bfloat16, non_blocking=True) for f in fr]) if len(fr) > 1 else \
            fr[0].to(device=model.device, dtype=torch.bfloat16, non_blocking=True)
        return model.prefetch(pixel_values=pv, n_clips=len(group)) if ahead else pv

    def row(t):
        return t.reshape(-1) if t.dim() <= 1 or t.shape[0] != 1 else t[0].reshape(-1)

    def enqueue(group, front):
        """ONE forward for the group: everything is enqueued, nothing is waited for."""
        ids = [row(it["input_ids"]) for it, _ in group]
        labels = [row(it["labels"]) for it, _ in group]
        masks = [row(it["attention_mask"]).bool() if it.get("attention_mask") is not None else torch.ones_like(i, dtype=torch.bool) for (it, _), i in zip(group, ids)]
        n = [int(i.numel()) for i in ids]
        nmax = max(n)
        pad = lambda t, v: F.pad(t.cpu(), (0, nmax - t.numel()), value=v)
        flags = None
        if all(it.get("image_flags") is not None for it, _ in group):
            flags = torch.cat([(it["image_flags"][0] if it["image_flags"].dim() == 3 else it["image_flags"]).reshape(-1, 1) for it, _ in group])
        motion = None                                      # (optional: items that carry a precomputed SlowFast feature [1, motion_dim])
        if all(it.get("motion_feature") is not None for it, _ in group):
            motion = torch.cat([it["motion_feature"].reshape(1, -1) for it, _ in group])
        out = model(mos=None, pixel_values=front, input_ids=torch.stack([pad(i, pad_id) for i in ids]),
                    attention_mask=torch.stack([pad(m, False) for m in masks]), image_flags=flags,
                    labels=torch.stack([pad(l, -100) for l in labels]), **({} if motion is None else {"motion_feature": motion}))
        return group, n, nmax, out

    def collect(run):
        """The group's results to the host - ONE synchronisation per group (the plain loop has one per clip: score1.item(), stage2_eval.py:938) - and
        out per item."""
        group, n, nmax, out = run
        logit = out["logit"].view(len(group), nmax - 1).cpu()
        label = out["label"].view(len(group), nmax - 1).cpu()
        score1 = out["score1"].cpu() if "score1" in out else None
        for b, (it, _) in enumerate(group):
            o = {"logit": logit[b, : n[b] - 1].clone(), "label": label[b, : n[b] - 1].clone()}
            if score1 is not None:
                o["score1"] = score1[b: b + 1].clone()
                mos = it.get("mos")
                o["loss"] = F.l1_loss(o["score1"], row(mos)[:1].to(o["score1"].dtype)) if mos is not None else None
            yield it, o

    it = groups()
    try:
        cur = next(it)
    except StopIteration:
        return
    front = start(cur)
    # Software pipeline, two deep: group g's forward is ENQUEUED before group g - 1's results are fetched, so the GPU already holds its next pass
    # while the host copies results back and the caller's loop body (token decoding, CSV rows) runs; the results of a group therefore come
    # out one group late - every item still comes out, in order.
    pending = None
    for nxt in it:
        nxt_front = start(nxt) if ahead else None          # enqueued BEFORE this group's InternLM2 pass: the two run side by side
        run = enqueue(cur, front)
        if pending is not None:
            yield from collect(pending)
        pending = run
        cur, front = nxt, nxt_front if ahead else start(nxt)
    run = enqueue(cur, front)
    if pending is not None:
        yield from collect(pending)
    yield from collect(run)


def score_dataset(items, model, tokenizer, k: int = 4, frames=None, group=None, output_file: Optional[str] = None, im_end_id: int = 92542):
    """The body of the reference's eval loop as ONE call (stage2_eval.py:908-972; stage1_eval.py:905-960 is the same body without the score):
    every item of ``items`` (this rank's share: pass ``shard(dataset, rank, world)`` wrapped in the driver's ``DataLoader(batch_size=1)``) scored
    through ``batched`` in groups of ``k``; per item the answer-token slice (:940-941) is decoded with ``tokenizer``, parsed into a level
    (:956-967) and appended as the driver's row ``[video_name, answer, output, mos, score1, level]`` (:970); the rows of all ranks of ``group``
    are put back into the set's order (``gather_rows``) and, on the caller's side, go through ``save_and_evaluate`` (:973) when ``output_file``
    is given or metrics are wanted.  Returns ``(rows, metrics)``; ``metrics`` is None when the set is empty.  A stage-1 model returns no ``score1``: its
    rows carry 0.0 in that column."""
    rows = []
    for item, out in batched(items, model, k=k, frames=frames):
        labels_row = item["labels"][0] if item["labels"].dim() == 2 else item["labels"]
        pred = answer_ids(labels_row, out["logit"], im_end_id=im_end_id)
        text = tokenizer.decode(pred)
        name = item["video_name"][0] if isinstance(item.get("video_name"), (list, tuple)) else item.get("video_name")
        answer = item["answer"][0] if isinstance(item.get("answer"), (list, tuple)) else item.get("answer", "")
        mos = float(item["mos"].reshape(-1)[0]) if item.get("mos") is not None else 0.0
        score1 = float(out["score1"].float().item()) if "score1" in out else 0.0
        rows.append([name, answer, text, mos, score1, parse_level(text)])
    rows = gather_rows(rows, group)
    return rows, (save_and_evaluate(rows, output_file) if rows else None)
