"""Full-size golden vectors, third set (round 3): a SECOND batch of the benched configuration - 4 clips x 8 frames x 448 px, N = 2177, the
same seeded weights as make_golden_8b.py / make_golden_8b_r3.py, inputs of seed 1 instead of bench.py's seed 0 - through the imported
REFERENCE (CPU eager path; internvl/model/internvl_chat_eval2/modeling_internvl_chat.py:306-488, stage2_eval.py:930-941) in fp32 and bf16,
so that the full-size parity statistics of tests/test_gpu_e2e.py do not rest on one batch.  Outputs only are recorded.

Run (build container only; ~50 GB of RAM, ~20 min on 8 cores):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_8b_r3b.py

Output: tests/golden/e2e_8b_r3b.pt (plain tensors / lists / dicts: loads with weights_only=True)
"""
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
from make_golden_8b import OVERRIDES, PLANT_SCALE, W_SEED, quiet, reference_dims  # noqa: E402
from make_golden_8b_r3 import BENCH_B, BENCH_T, run  # noqa: E402

SEED_B = 1


def main():
    llm, vis = reference_dims()
    out_path = os.path.join(HERE, "e2e_8b_r3b.pt")
    if os.environ.get("AIGV_GOLDEN_DRY"):                                        # script rehearsal at two layers each; writes to /tmp
        llm["num_hidden_layers"], vis["num_hidden_layers"], out_path = 2, 2, "/tmp/e2e_8b_r3b_dry.pt"
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
    print(f"reference model with the seeded weights ready in {time.time() - t0:.0f} s", flush=True)
    out = dict(llm_config=llm, vision_config=vis, w_seed=W_SEED, plant_scale=PLANT_SCALE, overrides=dict(OVERRIDES), cases={})
    out["cases"]["batch4/fp32"] = run(model, SlowFastStandIn, cfg, SEED_B, torch.float32, BENCH_B, BENCH_T)
    model = model.to(torch.bfloat16)                                             # exact: every value is a bf16 number already
    out["cases"]["batch4/bf16"] = run(model, SlowFastStandIn, cfg, SEED_B, torch.bfloat16, BENCH_B, BENCH_T)
    torch.save(out, out_path)
    print("wrote", out_path, flush=True)


if __name__ == "__main__":
    main()
