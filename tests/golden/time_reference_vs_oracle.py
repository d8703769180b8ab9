"""Wall time of the imported REFERENCE against this repo's CPU oracle on the same host, inputs and thread count
(the oracle is the reference's stand-in for the CPU baseline on the GPU box, where the reference cannot run;
the condition is that the two take the same time within +-10 %).  Build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/time_reference_vs_oracle.py   ->  profiles/r2_oracle_vs_reference_walltime.txt

Configuration: InternVL2-8B WIDTHS (the reference's config.json) at reduced depth (2 ViT + 2 LLM layers; the per-layer cost is what
the baseline scales by depth), one 8-frame 448x448 clip, N = 2177, bf16 and fp32, best of 3.
"""
import contextlib
import io
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_shims  # noqa: E402
import aigv_assessor_amd as pkg  # noqa: E402
from aigv_assessor_amd import synth  # noqa: E402
from make_golden_8b import reference_dims  # noqa: E402
from oracle import oracle as O  # noqa: E402


def best(fn, reps=3):
    fn()
    t = []
    for _ in range(reps):
        t0 = time.time()
        fn()
        t.append(time.time() - t0)
    return min(t)


def main():
    os.environ.setdefault("MASTER_PORT", "29579")
    llm, vis = reference_dims()
    llm["num_hidden_layers"], vis["num_hidden_layers"] = 2, 2
    cfg = pkg.InternVLChatConfig.from_dict(dict(vision_config=vis, llm_config=llm, force_image_size=448, select_layer=-1))
    m2, _m1, cfg2, SlowFastStandIn = ref_shims.install(llm, vis)
    lines = [f"# reference vs oracle wall time, {torch.get_num_threads()} threads, InternVL2-8B widths, 2 + 2 layers, 1 clip x 8 x 448 px, N = 2177"]
    for dt in (torch.bfloat16, torch.float32):
        sd = synth.make_state_dict(cfg, seed=7, dtype=dt, rich=True)
        with contextlib.redirect_stdout(io.StringIO()):
            rcfg = cfg2.InternVLChatConfig(select_layer=-1, force_image_size=448, downsample_ratio=0.5, template="internlm2-chat", ps_version="v2")
            model = m2.InternVLChatModel(rcfg).to(dt).eval()
        model.load_state_dict(sd, strict=True)
        toks = synth.canonical_tokens(cfg, 1, 8, seed=7)
        model.img_context_token_id = toks["img_context_token_id"]
        pv = synth.synthetic_frames(8, 448, seed=7, dtype=dt)
        motion = synth.synthetic_motion(1, 2304, seed=7, dtype=dt)
        SlowFastStandIn.feature = motion
        flags = torch.ones(8, 1, dtype=torch.long)

        def run_ref():
            with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
                return model(mos=torch.full((1,), 0.5, dtype=dt), pixel_values=pv, input_ids=toks["input_ids"], attention_mask=toks["attention_mask"],
                             image_flags=flags, labels=toks["labels"])

        def run_oracle():
            with torch.no_grad():
                return O.forward_eval(sd, cfg, pv, toks["input_ids"], toks["attention_mask"], flags, toks["labels"], motion,
                                      toks["img_context_token_id"], mos=torch.full((1,), 0.5, dtype=dt), stage=2)
        a, b = run_ref(), run_oracle()
        same = torch.equal(a["score1"], b["score1"]) and torch.equal(a["logit"], b["logit"])
        t_ref, t_or = best(run_ref), best(run_oracle)
        lines.append(f"{str(dt):16s} reference {t_ref:7.2f} s   oracle {t_or:7.2f} s   oracle / reference = {t_or / t_ref:.3f}   outputs identical: {same}")
        print(lines[-1], flush=True)
        del model, sd
    out = os.path.join(ROOT, "profiles", "r2_oracle_vs_reference_walltime.txt")
    with open(out, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("wrote", out)


if __name__ == "__main__":
    main()
