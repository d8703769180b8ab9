"""Beam search of ``generate(num_beams > 1, do_sample=False)``: host-side search logic over a model that is only seen through three callables.

The reference decodes through HF's ``language_model.generate(inputs_embeds=...)`` (modeling_internvl_chat.py:798-809), so ``num_beams > 1`` in a
generation config means transformers' beam search.  This module restates that published algorithm (transformers/generation/utils.py,
``GenerationMixin._beam_search`` and its helpers ``_get_top_k_continuations`` / ``_get_running_beams_for_next_iteration`` /
``_update_finished_beams`` / ``_check_early_stop_heuristic`` - the vectorised form of transformers >= 4.50) and is pinned token for token
against the INSTALLED transformers' ``generate`` (5.x here) on a small causal LM in tests/test_host.py.  The reference pins transformers
4.37.2, whose ``BeamSearchScorer`` runs the same search (scores, top-2k candidate rule, length penalty) but differs in two corners that
are NOT pinned here: with ``early_stopping=False`` its ``BeamHypotheses.is_done`` compares against the best of ALL 2k candidate scores
(end-token candidates included) where the vectorised heuristic below uses the best RUNNING beam, so stop points can differ in edge
cases; and it fills finished rows with ``pad_token_id`` even when that id is 0, where this code (like transformers 5.x) treats a pad id
of 0 as unset and fills with the end token.  The reference's shipped generation configs are ``num_beams=1`` (no ``num_beams`` anywhere in
its tree), so neither corner is reachable from its eval scripts.  With ``inputs_embeds``
HF's ``input_ids`` start empty, so the decoder prompt length is 0 here and every length below counts GENERATED tokens.

    first_logits  fp32 [B, V]: next-token logits behind the prompt (all beams of an item start from the same state)
    step(tok)     tok long [B, nb]: the token each running beam just took (after ``reorder``) -> fp32 logits [B, nb, V]
    reorder(par)  par long [B, nb]: running beam k of item b continues from what beam par[b, k] has cached
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import torch

NEG = -1.0e9      # transformers' "minus infinity" of the beam scores (kept as a finite number there, so kept here)


def _gather(x: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """x [B, K, ...], idx [B, M] -> x[b, idx[b, m], ...]"""
    ix = idx
    while ix.dim() < x.dim():
        ix = ix.unsqueeze(-1)
    return torch.gather(x, 1, ix.expand(*idx.shape, *x.shape[2:]))


def beam_search(first_logits: torch.Tensor, step: Callable[[torch.Tensor], torch.Tensor], reorder: Callable[[torch.Tensor], None],
                num_beams: int, max_new_tokens: int, eos_ids: Sequence[int] = (), pad_id: Optional[int] = None,
                length_penalty: float = 1.0, early_stopping=False, processors: Sequence[Callable] = ()) -> torch.Tensor:
    """Best hypothesis per item: long [B, L] of NEW tokens, L = the longest returned hypothesis (an end token included), shorter ones
    filled with ``pad_id`` (or the first end token when no pad id is set, as HF does)."""
    if num_beams < 2:
        raise ValueError("beam_search needs num_beams >= 2")
    if max_new_tokens < 1:
        raise ValueError("max_new_tokens must be positive")
    if early_stopping not in (True, False, "never"):
        raise ValueError("early_stopping must be True, False or 'never'")
    dev = first_logits.device
    B, V = first_logits.shape
    nb, L = num_beams, max_new_tokens
    eos = torch.tensor(list(eos_ids), dtype=torch.long, device=dev) if len(eos_ids) else None
    keep = max(2, 1 + len(eos_ids)) * nb                        # candidates kept per item and step
    fill = (pad_id if pad_id else int(eos_ids[0])) if len(eos_ids) else -1
    top_mask = torch.arange(keep, device=dev) < nb

    run_seq = torch.full((B, nb, L), fill, dtype=torch.long, device=dev)      # running beams: tokens so far
    run_score = torch.zeros((B, nb), dtype=torch.float32, device=dev)
    run_score[:, 1:] = NEG                                                    # step 0 expands beam 0 only (the others are its copies)
    fin_seq = run_seq.clone()                                                 # finished hypotheses, best first
    fin_score = torch.full((B, nb), NEG, dtype=torch.float32, device=dev)
    fin_len = torch.zeros((B, nb), dtype=torch.long, device=dev)
    fin_done = torch.zeros((B, nb), dtype=torch.bool, device=dev)
    can_improve = torch.ones((B, 1), dtype=torch.bool, device=dev)            # "early-stop heuristic unsatisfied"

    cur = 0
    logits = first_logits.float()[:, None, :].expand(B, nb, V)
    while True:
        logp = torch.log_softmax(logits.reshape(B * nb, V).float(), dim=-1)
        hist = run_seq[:, :, :cur].reshape(B * nb, cur)
        for proc in processors:                                               # HF applies the logits processors to the log-probabilities
            logp = proc(hist, logp)
        acc = (logp.view(B, nb, V) + run_score[:, :, None]).view(B, nb * V)
        # the `keep` best continuations of every item, over all of its beams
        top_score, top_idx = torch.topk(acc, k=keep)
        top_beam = top_idx // V
        top_tok = top_idx % V
        cand_seq = _gather(run_seq, top_beam)
        cand_seq[:, :, cur] = top_tok
        hit = torch.zeros_like(top_tok, dtype=torch.bool) if eos is None else torch.isin(top_tok, eos)
        hit = hit | (cur + 1 >= L)                                            # stopping criteria: an end token, or the length limit
        # running beams of the next step: the best candidates that did not stop
        alive_score = top_score + hit.float() * NEG
        nxt = torch.topk(alive_score, k=nb)[1]
        run_seq = _gather(cand_seq, nxt)
        run_score = _gather(alive_score, nxt)
        parent = _gather(top_beam, nxt)
        # finished hypotheses: only a candidate among the item's first nb may finish; length-normalised score
        just = hit & top_mask[None, :]
        fscore = top_score / float((cur + 1) ** length_penalty)
        fscore = fscore + (fin_done.all(dim=-1, keepdim=True) & (early_stopping is True)).float() * NEG
        fscore = fscore + (~can_improve).float() * NEG
        fscore = fscore + (~just).float() * NEG
        m_seq = torch.cat((fin_seq, cand_seq), dim=1)
        m_score = torch.cat((fin_score, fscore), dim=1)
        m_len = torch.cat((fin_len, torch.full_like(top_tok, cur + 1)), dim=1)
        m_done = torch.cat((fin_done, just), dim=1)
        best = torch.topk(m_score, k=nb)[1]
        fin_seq, fin_score, fin_len, fin_done = _gather(m_seq, best), _gather(m_score, best), _gather(m_len, best), _gather(m_done, best)

        cur += 1
        # can a running beam still beat the worst finished hypothesis of its item?
        hyp_len = L if (early_stopping == "never" and length_penalty > 0.0) else cur
        best_running = run_score[:, :1] / float(hyp_len ** length_penalty)
        worst_finished = torch.where(fin_done, fin_score.min(dim=1, keepdim=True)[0], torch.full_like(fin_score, NEG))
        can_improve = can_improve & (best_running > worst_finished).any(dim=-1, keepdim=True)
        go_on = bool(can_improve.any()) and not (bool(fin_done.all()) and early_stopping is True) and not bool(hit.all())
        if not go_on:
            break
        reorder(parent)
        logits = step(run_seq[:, :, cur - 1])

    out_len = max(1, int(fin_len[:, 0].max()))
    return fin_seq[:, 0, :out_len]


def parents_to_slots(parent: torch.Tensor, slot_of: Callable[[int, int], int]) -> List[int]:
    """[B, nb] parents -> for every cache slot (any layout given by ``slot_of(b, k)``) the slot it continues from."""
    B, nb = parent.shape
    par = parent.tolist()
    src = [0] * (B * nb)
    for b in range(B):
        for k in range(nb):
            src[slot_of(b, k)] = slot_of(b, par[b][k])
    return src
