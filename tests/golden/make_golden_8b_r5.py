"""Full-size golden vectors, fifth set (round 5): a SAMPLE large enough to read the parity bars from.

Round 4 measured the reference's bf16 CPU eager pass against ITSELF on five pairs (host thread counts 8 vs 4 / 1).  This script widens
both samples, with the same seeded InternVL2-8B-size weights as make_golden_8b*.py (W_SEED, OVERRIDES):

* ``--phase threads --threads N [N ...]``: all 13 clips the reference was recorded on so far - the five one-clip input seeds of
  e2e_8b_full.pt (201-205, scored alone) and the two batches of four of e2e_8b_r3.pt / e2e_8b_r3b.pt (input seeds 0 and 1) - re-scored
  in bf16 under ``torch.set_num_threads(N)``; with N in {1, 2, 4} against the recorded 8-thread values that is 39 reference-vs-itself
  pairs (oneDNN / MKL choose blockings and K splits by thread count: another fp32 summation order of the same arithmetic).
  ``one/201/t8`` is re-run first and must reproduce the recorded fixture bit for bit.
* ``--phase new``: six more batches of the benched shape (4 clips x 8 frames x 448 px, N = 2177; input seeds 2-7) in bf16 with 8
  threads: 24 more reference-pinned clips (37 in all).
* ``--phase fp32``: the same six batches through the fp32 model (the "truth" column of tests/manual/parity_stats.py).
* ``--merge``: the per-phase files -> tests/golden/e2e_8b_r5.pt.

(reference: internvl/model/internvl_chat_eval2/modeling_internvl_chat.py:306-488, internvl/model/internlm2/modeling_internlm2.py:407-424,
internvl/train/internvl/eval/stage2_eval.py:908-941.)  Outputs only are recorded; every phase saves after every case, so an interrupted
run keeps what it has.

Run (build container only; bf16 phases ~17 GB of RAM each, fp32 ~34 GB; a batch of four takes ~3 min at 8 threads, ~25 min at 1):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_8b_r5.py --phase new
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_8b_r5.py --phase threads --threads 4 2
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_8b_r5.py --phase threads --threads 1
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_8b_r5.py --phase fp32
    python tests/golden/make_golden_8b_r5.py --merge

Output: tests/golden/e2e_8b_r5.pt (plain tensors / lists / dicts: loads with weights_only=True)
"""
import argparse
import glob
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

ONE_SEEDS = (201, 202, 203, 204, 205)
OLD_BATCH_SEEDS = (0, 1)
NEW_BATCH_SEEDS = (2, 3, 4, 5, 6, 7)
B4, T = 4, 8


def score(model, SlowFastStandIn, cfg, synth, quiet, seed, B, dt, threads):
    """One pass of the reference over the B clips of input seed ``seed`` with ``threads`` host threads."""
    toks = synth.canonical_tokens(cfg, B, T, seed=seed)
    pv = synth.synthetic_frames(B * T, 448, seed=seed, dtype=dt)
    motion = synth.synthetic_motion(B, 2304, seed=seed, dtype=dt)
    model.img_context_token_id = toks["img_context_token_id"]
    SlowFastStandIn.feature = motion
    grabbed = {}
    hooks = [model.language_model.register_forward_hook(lambda m, i, o: grabbed.__setitem__("logits", o.logits)),
             model.language_model.model.register_forward_hook(lambda m, i, o: grabbed.__setitem__("hidden", o.last_hidden_state))]
    torch.set_num_threads(threads)
    t0 = time.time()
    with torch.no_grad(), quiet():
        out = model(mos=torch.full((B,), 0.5, dtype=dt), pixel_values=pv, input_ids=toks["input_ids"], attention_mask=toks["attention_mask"],
                    image_flags=torch.ones(B * T, 1, dtype=torch.long), labels=toks["labels"])
    for h in hooks:
        h.remove()
    want = out["label"] != -100
    V = grabbed["logits"].shape[-1]
    rows = grabbed["logits"][..., :-1, :].reshape(-1, V)[want].float()
    top_v, top_i = rows.topk(4, dim=-1)
    rec = dict(seed=seed, B=B, T=T, dtype=str(dt), threads=threads, score1=out["score1"].clone(), logit=out["logit"][want].clone(),
               answer_rows=want.nonzero().flatten().clone(), top_values=top_v.clone(), top_ids=top_i.clone(),
               hidden_m4=grabbed["hidden"][:, -4, :].clone(), seconds=time.time() - t0)
    print(f"  seed {seed} B={B} {dt} threads {threads}: score1 {out['score1'].float().tolist()} argmax {out['logit'][want].tolist()} "
          f"({rec['seconds']:.0f} s)", flush=True)
    return rec


def build(dt):
    import ref_shims
    import aigv_assessor_amd as pkg
    from aigv_assessor_amd import synth
    from make_golden_8b import OVERRIDES, PLANT_SCALE, W_SEED, quiet, reference_dims
    llm, vis = reference_dims()
    if os.environ.get("AIGV_GOLDEN_DRY"):                                        # script rehearsal at two layers each
        llm["num_hidden_layers"], vis["num_hidden_layers"] = 2, 2
    cfg = pkg.InternVLChatConfig.from_dict(dict(vision_config=vis, llm_config=llm, force_image_size=448, select_layer=-1))
    m2, _m1, cfg2, SlowFastStandIn = ref_shims.install(llm, vis)
    t0 = time.time()
    with quiet():
        rcfg = cfg2.InternVLChatConfig(select_layer=-1, force_image_size=448, downsample_ratio=0.5, template="internlm2-chat", ps_version="v2")
        model = m2.InternVLChatModel(rcfg).eval()
    sd = synth.make_state_dict(cfg, seed=W_SEED, rich=True)
    for k, v in OVERRIDES.items():
        sd[k] = torch.full_like(sd[k], v)
    model.load_state_dict(sd, strict=True)
    del sd
    if dt == torch.bfloat16:
        model = model.to(torch.bfloat16)                                         # exact: every value is a bf16 number already
    print(f"reference model ({dt}) with the seeded weights ready in {time.time() - t0:.0f} s", flush=True)
    head = dict(llm_config=llm, vision_config=vis, w_seed=W_SEED, plant_scale=PLANT_SCALE, overrides=dict(OVERRIDES),
                host=dict(torch=str(torch.__version__), cpus=os.cpu_count()))
    return model, SlowFastStandIn, cfg, synth, quiet, head


def part_path(tag):
    d = "/tmp" if os.environ.get("AIGV_GOLDEN_DRY") else HERE
    return os.path.join(d, f"e2e_8b_r5.part_{tag}.pt")


def phase_threads(threads):
    model, SF, cfg, synth, quiet, head = build(torch.bfloat16)
    out = dict(head, cases={})
    path = part_path("threads_" + "_".join(str(t) for t in threads))
    if not os.environ.get("AIGV_GOLDEN_DRY"):                                    # the recorded 8-thread pass must come out again bit for bit
        old = torch.load(os.path.join(HERE, "e2e_8b_full.pt"), weights_only=True)["cases"]["bf16/201"]
        r = score(model, SF, cfg, synth, quiet, 201, 1, torch.bfloat16, 8)
        assert torch.equal(old["score1"], r["score1"]) and torch.equal(old["logit"], r["logit"]), "the recorded 8-thread pass is not reproduced"
        print("  one/201/t8 reproduces e2e_8b_full.pt", flush=True)
        out["cases"]["one/201/t8"] = r
    for t in threads:
        for s in OLD_BATCH_SEEDS:
            out["cases"][f"batch4/seed{s}/t{t}"] = score(model, SF, cfg, synth, quiet, s, B4, torch.bfloat16, t)
            torch.save(out, path)
        for s in ONE_SEEDS:
            out["cases"][f"one/{s}/t{t}"] = score(model, SF, cfg, synth, quiet, s, 1, torch.bfloat16, t)
            torch.save(out, path)
    print("wrote", path, flush=True)


def phase_new(dt):
    model, SF, cfg, synth, quiet, head = build(dt)
    out = dict(head, cases={})
    name = "bf16" if dt == torch.bfloat16 else "fp32"
    path = part_path("new_" + name)
    for s in NEW_BATCH_SEEDS:
        out["cases"][f"batch4/seed{s}/{name}"] = score(model, SF, cfg, synth, quiet, s, B4, dt, 8)
        torch.save(out, path)
    print("wrote", path, flush=True)


def merge():
    parts = sorted(glob.glob(part_path("*")))
    assert parts, "no part files"
    out = None
    for p in parts:
        d = torch.load(p, weights_only=True)
        if out is None:
            out = {k: v for k, v in d.items() if k != "cases"}
            out["cases"] = {}
        for k in ("llm_config", "vision_config", "w_seed", "overrides"):
            assert out[k] == d[k], (p, k)
        for k, v in d["cases"].items():
            if k in out["cases"]:                                                # (one/201/t8, the reproduction check, is re-run by every threads phase)
                assert torch.equal(out["cases"][k]["score1"], v["score1"]) and torch.equal(out["cases"][k]["logit"], v["logit"]), k
                continue
            out["cases"][k] = v
    dst = os.path.join(HERE, "e2e_8b_r5.pt")
    torch.save(out, dst)
    print("merged", [os.path.basename(p) for p in parts], "->", dst, f"({len(out['cases'])} cases)")
    for k in sorted(out["cases"]):
        c = out["cases"][k]
        print(f"  {k:24s} score1 {[round(x, 4) for x in c['score1'].float().tolist()]} ({c['seconds']:.0f} s)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--phase", choices=("threads", "new", "fp32"))
    ap.add_argument("--threads", type=int, nargs="+", default=[4, 2, 1])
    ap.add_argument("--merge", action="store_true")
    a = ap.parse_args()
    if a.merge:
        return merge()
    if a.phase == "threads":
        return phase_threads(a.threads)
    if a.phase == "new":
        return phase_new(torch.bfloat16)
    if a.phase == "fp32":
        return phase_new(torch.float32)
    ap.error("nothing to do")


if __name__ == "__main__":
    main()
