"""Full-size parity statistics on an MI355X: every clip the imported reference was recorded on, scored under

    attention numerics {reference score rounding, fp32 scores}  [x GEMM dispatch modes {0 row plans, 1 128 kernel, 2 256 kernel} with --modes]

next to the reference's own spread against itself.  Round 5 sample:

* 37 reference-pinned clips: five one-clip seeds (tests/golden/e2e_8b_full.pt), the two batches of four of rounds 3 (e2e_8b_r3.pt / _r3b.pt) and six
  more batches of the benched shape (e2e_8b_r5.pt, input seeds 2-7), each with the reference's bf16 score and - where recorded - its fp32 score;
* 39 reference-vs-itself pairs: the first 13 clips re-scored by the reference under torch.set_num_threads(1 / 2 / 4) against its 8-thread pass
  (e2e_8b_r5.pt; another fp32 summation order of the same arithmetic) + the five pairs of round 4 (e2e_8b_r4_self.pt).

Every (numerics, mode) pair is a CORRECT evaluation of the same arithmetic definition up to fp32 summation order (and, for the numerics, up to where
the score matrix rounds).  Printed per arm: mean / 95th percentile / max of |hip - ref bf16| and of |hip - ref fp32| in bf16 ulps of the score, the
number of clips with the identical bf16 score, level tokens differing; then the same statistics of the reference against itself.

Round 6: the TASK-LEVEL agreement - SRCC / PLCC / KRCC, the reference's own quality metric (stage2_eval.py:652-688) - of the
HIP scores against the reference's bf16 and fp32 scores over the 37 clips, next to the same statistics of the reference's bf16 pass against its fp32
pass and of the reference against itself under other host thread counts; for bf16 and, with --fp8, for set_precision("fp8").

    python tests/manual/parity_stats.py [out.json] [--modes] [--fp8]
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import aigv_assessor_amd as pkg  # noqa: E402
from aigv_assessor_amd import synth  # noqa: E402
from aigv_assessor_amd.modeling import InternVLChatModel  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")


def ulp(x):
    return 2.0 ** (torch.tensor(abs(float(x))).clamp_min(1e-30).log2().floor().item() - 7)


def stats(v):
    t = torch.tensor(v, dtype=torch.float64)
    return dict(n=len(v), mean=float(t.mean()), p95=float(t.quantile(0.95)), max=float(t.max()))


def fmt(s):
    return f"mean {s['mean']:.2f}  p95 {s['p95']:.1f}  max {s['max']:.1f}  (n = {s['n']})"


def corr(a, b):
    from scipy.stats import kendalltau, pearsonr, spearmanr
    return dict(srcc=float(spearmanr(a, b)[0]), plcc=float(pearsonr(a, b)[0]), krcc=float(kendalltau(a, b)[0]), n=len(a))


def fmtc(c):
    return f"SRCC {c['srcc']:.4f}  PLCC {c['plcc']:.4f}  KRCC {c['krcc']:.4f}  (n = {c['n']})"


def reference_cases():
    """[(name, B, input seed, ref bf16 scores [B], ref fp32 scores [B] or None, ref bf16 answer tokens, answer rows)]"""
    g = torch.load(os.path.join(G, "e2e_8b_full.pt"), weights_only=True)
    cases = []
    for seed in (201, 202, 203, 204, 205):
        a, b = g["cases"][f"bf16/{seed}"], g["cases"][f"fp32/{seed}"]
        cases.append((f"one/{seed}", 1, seed, a["score1"].float(), b["score1"].float(), a["logit"], a["answer_rows"]))
    for f in ("e2e_8b_r3.pt", "e2e_8b_r3b.pt"):
        gg = torch.load(os.path.join(G, f), weights_only=True)["cases"]
        a, b = gg["batch4/bf16"], gg["batch4/fp32"]
        cases.append((f"batch4/seed{a['seed']}", 4, a["seed"], a["score1"].float(), b["score1"].float(), a["logit"], a["answer_rows"]))
    r5 = os.path.join(G, "e2e_8b_r5.pt")
    if os.path.exists(r5):
        c5 = torch.load(r5, weights_only=True)["cases"]
        for seed in range(2, 8):
            a = c5.get(f"batch4/seed{seed}/bf16")
            if a is None:
                continue
            b = c5.get(f"batch4/seed{seed}/fp32")
            cases.append((f"batch4/seed{seed}", 4, seed, a["score1"].float(), None if b is None else b["score1"].float(), a["logit"], a["answer_rows"]))
    return g, cases


def reference_self_pairs(cases):
    """|ref bf16 under N threads - ref bf16 under 8 threads| in ulps, per clip, for every recorded thread count; level tokens flipped."""
    out, flips, rows = {}, 0, 0
    base = {name: (r16, rlog) for name, _B, _s, r16, _r32, rlog, _rows in cases}
    r5 = os.path.join(G, "e2e_8b_r5.pt")
    if os.path.exists(r5):
        c5 = torch.load(r5, weights_only=True)["cases"]
        for key, c in c5.items():
            parts = key.split("/")
            if not parts[-1].startswith("t") or parts[-1] == "t8":
                continue
            name = "/".join(parts[:-1])
            if name not in base:
                continue
            r16, rlog = base[name]
            s = c["score1"].float()
            out.setdefault(parts[-1], []).extend(abs(float(s[i] - r16[i])) / ulp(r16[i]) for i in range(len(s)))
            flips += int((c["logit"] != rlog).sum()); rows += int(rlog.numel())
    r4 = os.path.join(G, "e2e_8b_r4_self.pt")
    if os.path.exists(r4):
        c = torch.load(r4, weights_only=True)["cases"]
        a, b = c["batch4/seed0/t8"]["score1"].float(), c["batch4/seed0/t4"]["score1"].float()
        out.setdefault("r4:t4", []).extend(abs(float(a[i] - b[i])) / ulp(a[i]) for i in range(4))
        out.setdefault("r4:t1", []).append(abs(float(c["alone/seed0/clip0/t1"]["score1"].float()[0] - c["alone/seed0/clip0/t8"]["score1"].float()[0])) / ulp(a[0]))
    return out, flips, rows


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    modes = (0, 1, 2) if "--modes" in sys.argv else (0,)
    precisions = ("bf16", "fp8") if "--fp8" in sys.argv else ("bf16",)
    g, cases = reference_cases()
    cfg = pkg.InternVLChatConfig.from_dict(dict(vision_config=g["vision_config"], llm_config=g["llm_config"], force_image_size=448, select_layer=-1))
    dev = torch.device("cuda", 0)
    model = InternVLChatModel(cfg, device=dev, max_clips=4, max_frames=32, max_tokens=4 * synth.canonical_len(cfg, 8))
    sd = synth.make_state_dict(cfg, seed=g["w_seed"], rich=True)
    for k, v in g.get("overrides", {}).items():
        sd[k] = torch.full_like(sd[k], v)
    model.load_state_dict(sd)
    del sd
    model.eval()
    n_clips = sum(c[1] for c in cases)
    print(f"{n_clips} reference-pinned clips in {len(cases)} recorded passes", flush=True)
    out = {}
    for precision in precisions:
      model.set_precision(precision)
      for numerics in ("reference", "fp32"):
        model.set_attention_numerics(numerics)
        for mode in modes:
            model.set_gemm_mode(mode)
            d16, d32, lev, nlev, per_clip, s_hip, s16, s_hip32, s32 = [], [], 0, 0, [], [], [], [], []
            for name, B, seed, r16, r32, rlog, rows in cases:
                toks = synth.canonical_tokens(cfg, B, 8, seed=seed)
                model.img_context_token_id = toks["img_context_token_id"]
                o = model(mos=None, pixel_values=synth.synthetic_frames(B * 8, 448, seed=seed).to(dev), input_ids=toks["input_ids"],
                          attention_mask=toks["attention_mask"], image_flags=torch.ones(B * 8, 1, dtype=torch.long), labels=toks["labels"],
                          motion_feature=synth.synthetic_motion(B, cfg.motion_dim, seed=seed).to(dev))
                torch.cuda.synchronize()
                hip = o["score1"].float().cpu()
                for i in range(B):
                    u = ulp(r16[i])
                    d16.append(abs(float(hip[i] - r16[i])) / u)
                    s_hip.append(float(hip[i])); s16.append(float(r16[i]))
                    if r32 is not None:
                        d32.append(abs(float(hip[i] - r32[i])) / u)
                        s_hip32.append(float(hip[i])); s32.append(float(r32[i]))
                    per_clip.append(round(d16[-1], 1))
                got = o["logit"].cpu()[rows]
                lev += int((got != rlog).sum())
                nlev += int(rlog.numel())
            key = f"{precision}/{numerics}/gemm{mode}"
            out[key] = dict(vs_ref_bf16=stats(d16), vs_ref_fp32=stats(d32), identical=int(sum(1 for x in d16 if x == 0)),
                            level_tokens_differing=lev, level_rows=nlev, per_clip_ulps=per_clip, corr_vs_ref_bf16=corr(s_hip, s16), corr_vs_ref_fp32=corr(s_hip32, s32),
                            scores=s_hip)
            print(f"{key:22s} |hip - ref bf16| {fmt(out[key]['vs_ref_bf16'])}, identical {out[key]['identical']}/{n_clips};  |hip - ref fp32| {fmt(out[key]['vs_ref_fp32'])};  "
                  f"level tokens differing {lev}/{nlev}", flush=True)
            print(f"{'':22s} task level: hip vs ref bf16 {fmtc(out[key]['corr_vs_ref_bf16'])};  hip vs ref fp32 {fmtc(out[key]['corr_vs_ref_fp32'])}", flush=True)
            print(f"{'':22s} per clip {per_clip}", flush=True)
    model.set_precision("bf16")
    model.set_gemm_mode(-1)
    # the reference's own two precisions, and its own spread against itself
    r = [abs(float(r16[i] - r32[i])) / ulp(r16[i]) for _n, B, _s, r16, r32, *_x in cases if r32 is not None for i in range(B)]
    out["ref_bf16_vs_ref_fp32"] = stats(r)
    print(f"reference bf16 vs reference fp32: {fmt(out['ref_bf16_vs_ref_fp32'])}")
    a16 = [float(r16[i]) for _n, B, _s, r16, r32, *_x in cases if r32 is not None for i in range(B)]
    a32 = [float(r32[i]) for _n, B, _s, r16, r32, *_x in cases if r32 is not None for i in range(B)]
    out["corr_ref_bf16_vs_ref_fp32"] = corr(a16, a32)
    print(f"task level: reference bf16 vs reference fp32 {fmtc(out['corr_ref_bf16_vs_ref_fp32'])}")
    # the reference against itself, task level: its bf16 scores under 1 / 2 / 4 host threads against its 8-thread pass (the 13 re-scored clips)
    c5p = os.path.join(G, "e2e_8b_r5.pt")
    if os.path.exists(c5p):
        c5 = torch.load(c5p, weights_only=True)["cases"]
        px, py = [], []
        for t in ("t1", "t2", "t4"):
            x, y = [], []
            for name, B, _s, r16, *_x in cases:
                if f"{name}/{t}" in c5:
                    x += c5[f"{name}/{t}"]["score1"].float().tolist(); y += r16.tolist()
            if x:
                out[f"corr_ref_{t}_vs_t8"] = corr(x, y)
                print(f"task level: reference {t} vs its 8-thread pass {fmtc(out[f'corr_ref_{t}_vs_t8'])}")
                px += x; py += y
        if px:
            out["corr_ref_vs_itself_pooled"] = corr(px, py)
            print(f"task level: reference vs itself, pooled {fmtc(out['corr_ref_vs_itself_pooled'])}")
    pairs, flips, rows = reference_self_pairs(cases)
    allp = [x for k, v in pairs.items() for x in v]
    for k in sorted(pairs):
        print(f"reference vs itself, {k:6s} against 8 host threads: {fmt(stats(pairs[k]))}")
    if allp:
        out["reference_vs_itself"] = dict(all=stats(allp), by_threads={k: stats(v) for k, v in pairs.items()}, level_tokens_flipped=flips, level_rows=rows,
                                          identical=int(sum(1 for x in allp if x == 0)))
        print(f"reference vs itself, all pairs: {fmt(out['reference_vs_itself']['all'])}, identical {out['reference_vs_itself']['identical']}/{len(allp)}; "
              f"level tokens flipped {flips}/{rows}")
    if args:
        json.dump(out, open(args[0], "w"), indent=1)


if __name__ == "__main__":
    main()
