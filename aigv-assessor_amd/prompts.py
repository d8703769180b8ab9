"""Host-side input building of the eval drivers: frame sampling and prompt -> ids -> label mask.

Deterministic integer / string code, restated from the reference and pinned to fixtures recorded from the reference's own
functions (tests/golden/make_host_golden.py -> tests/golden/host_inputs.pt):

* ``get_index``        stage2_eval.py:429-441   segment-centre frame indices of ``load_video``
* ``video_prompt``     stage2_eval.py:465-481   "<video>\\n" -> "Frame1: <image>\\n...FrameT: <image>\\nMotion Feature: <image>"
* ``build_inputs``     stage2_eval.py:488-498 + internvl/train/dataset.py:595-682 (``preprocess_internlm``): template join,
                       <image> expansion with ``num_image_tokens = [256] * T + [1]``, tokenisation, labels = the assistant's
                       answer tokens + the closing <|im_end|>, everything else -100.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch

from .conversation import get_conv_template

IGNORE_TOKEN_ID = -100          # transformers LabelSmoother.ignore_index (dataset.py:3,34)
IMG_START_TOKEN, IMG_END_TOKEN, IMG_CONTEXT_TOKEN = "<img>", "</img>", "<IMG_CONTEXT>"     # internvl/train/constants.py:1-3


def get_index(bound: Optional[Tuple[float, float]], fps: float, max_frame: int, first_idx: int = 0, num_segments: int = 8) -> List[int]:
    """stage2_eval.py:429-441: centres of ``num_segments`` equal segments of [start, end] (numpy's round = half to even,
    which Python's ``round`` shares; ``int()`` truncates)."""
    start, end = (bound[0], bound[1]) if bound else (-100000, 100000)
    start_idx = max(first_idx, round(start * fps))
    end_idx = min(round(end * fps), max_frame)
    seg = float(end_idx - start_idx) / num_segments
    return [int(start_idx + (seg / 2) + round(seg * i)) for i in range(num_segments)]


def video_prompt(question: str, n_frames: int) -> str:
    """stage2_eval.py:465-481: the user turn of a video sample."""
    if "<video>" not in question:
        question = "<video>\n" + question
    tokens = "\n".join(f"Frame{i + 1}: <image>" for i in range(n_frames)) + "\nMotion Feature: <image>"
    return question.replace("<video>\n", tokens)


def build_inputs(tokenizer, question: str, answer: str, n_frames: int, num_image_token: int = 256,
                 template: str = "internlm2-chat", system_message: Optional[str] = None) -> Dict[str, torch.Tensor]:
    """One stage-2 eval sample: ``input_ids`` / ``labels`` / ``attention_mask`` [N] exactly as ``video_get_item`` +
    ``preprocess_internlm`` build them (un-padded: the eval scripts run with group_by_length, dataset.py:636)."""
    conv = get_conv_template(template)
    if system_message is not None:
        conv.system_message = system_message
    conv.append_message(conv.roles[0], video_prompt(question, n_frames).strip())
    conv.append_message(conv.roles[1], answer.strip())
    text = conv.get_prompt()
    counts = [num_image_token] * n_frames + [1]                     # the 9th placeholder is the motion token (stage2_eval.py:493-494)
    for n in counts:
        text = text.replace("<image>", f"{IMG_START_TOKEN}{IMG_CONTEXT_TOKEN * n}{IMG_END_TOKEN}", 1)
    ids = tokenizer(text, return_tensors="pt", padding=False, max_length=tokenizer.model_max_length, truncation=True).input_ids[0]
    labels = ids.clone()
    # dataset.py:643-666: <s> and everything up to and including the assistant role string are ignored; the answer and its
    # <|im_end|> are kept; lengths are measured by re-tokenising the pieces (minus the <s> each call prepends)
    parts = text.split(conv.roles[1])
    cur = 1
    labels[:cur] = IGNORE_TOKEN_ID
    n0 = len(tokenizer(parts[0] + conv.roles[1]).input_ids) - 1
    labels[cur:cur + n0] = IGNORE_TOKEN_ID
    cur += n0
    for mid in parts[1:-1]:                                            # earlier assistant turns of a multi-turn sample
        p1, p2 = mid.split(conv.roles[0])
        cur += len(tokenizer(p1).input_ids) - 1
        n = len(tokenizer(conv.roles[0] + p2 + conv.roles[1]).input_ids) - 1
        labels[cur:cur + n] = IGNORE_TOKEN_ID
        cur += n
    cur += len(tokenizer(parts[-1]).input_ids) - 1
    labels[cur:] = IGNORE_TOKEN_ID
    if cur < tokenizer.model_max_length and cur != int(ids.ne(tokenizer.pad_token_id).sum()):
        labels[:] = IGNORE_TOKEN_ID                                    # the reference's "tokenization mismatch" fallback (:671-675)
    return {"input_ids": ids, "labels": labels, "attention_mask": ids.ne(tokenizer.pad_token_id)}


def batch_inputs(samples: Sequence[Dict[str, torch.Tensor]], pad_token_id: int = 2) -> Dict[str, torch.Tensor]:
    """Right-pad several samples to one [B, N] batch the way the reference's collator does (internvl/patch/pad_data_collator.py:
    ids with the pad id, labels with -100); the scorer strips the padding again (packed layout)."""
    n = max(int(s["input_ids"].numel()) for s in samples)
    ids = torch.full((len(samples), n), pad_token_id, dtype=torch.long)
    labels = torch.full((len(samples), n), IGNORE_TOKEN_ID, dtype=torch.long)
    mask = torch.zeros((len(samples), n), dtype=torch.bool)
    for i, s in enumerate(samples):
        k = int(s["input_ids"].numel())
        ids[i, :k], labels[i, :k], mask[i, :k] = s["input_ids"], s["labels"], True
    return {"input_ids": ids, "labels": labels, "attention_mask": mask}
