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

