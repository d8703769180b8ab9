"""Golden vectors for decoding PAST max_position_embeddings under dynamic-NTK rope scaling, recorded from the REFERENCE's InternLM2.

Real checkpoints ship rope_scaling = {"type": "dynamic", "factor": 2.0}.  The reference's rotary module rebuilds its cos / sin tables
with the base of the current kv_seq_len whenever that exceeds what it has cached (internvl/model/internlm2/modeling_internlm2.py:187-194,
227-243): during a KV-cache decode that is EVERY step past max_position_embeddings, and only the new token's q / k are rotated with the new
tables - cached keys keep the base they were written with.  This script runs the reference's own cache path (InternLM2ForCausalLM,
prepare_inputs_for_generation: LM:1126-1163) on a small seeded decoder whose limit is 32 positions:

* ``cross``: two prompts of 24 tokens, 72 new tokens - the decode crosses the limit at its 9th step and ends at three times the limit (the
  rotary base is then 5.1x the plain one);
* ``beyond``: one prompt of 80 tokens (the prompt pass itself is rescaled), 16 new tokens.

Outputs only are recorded (tokens, the top-2 logit gap of every step, the rotary tables the module ended with); weights come from
``aigv_assessor_amd.synth.make_state_dict`` with the recorded seed.  Run (build container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_ntk.py

Output: tests/golden/ntk_decode.pt
"""
import contextlib
import copy
import io
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

MAX_POS = 32
LLM = dict(hidden_size=512, intermediate_size=768, num_attention_heads=4, num_key_value_heads=2, num_hidden_layers=2, vocab_size=1024,
           rms_norm_eps=1e-5, rope_theta=1000000, max_position_embeddings=MAX_POS, rope_scaling={"factor": 2.0, "type": "dynamic"}, bias=False,
           hidden_act="silu", attn_implementation="eager", pad_token_id=2)
SEED = 77
CASES = {"cross": dict(b=2, prompt=24, new=72), "beyond": dict(b=1, prompt=80, new=16)}


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def prompt_ids(b, n, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(3, LLM["vocab_size"], (b, n), generator=g)


def main():
    import ref_shims
    import aigv_assessor_amd as pkg
    from aigv_assessor_amd import synth
    from make_golden import E2E_VIS
    ref_shims.install(dict(copy.deepcopy(LLM), architectures=["InternLM2ForCausalLM"]), E2E_VIS)
    import internvl.model.internlm2.modeling_internlm2 as rlm
    from internvl.model.internlm2.configuration_internlm2 import InternLM2Config as RLmCfg

    cfg = pkg.tiny(llm_hidden=LLM["hidden_size"], llm_heads=4, llm_kv_heads=2, llm_layers=2, llm_inter=LLM["intermediate_size"], vocab=LLM["vocab_size"])
    cfg.llm_config.max_position_embeddings = MAX_POS
    cfg.llm_config.rope_scaling = dict(LLM["rope_scaling"])
    sd = synth.make_state_dict(cfg, seed=SEED, rich=True)
    out = dict(llm_config=copy.deepcopy(LLM), seed=SEED, cases={})
    for name, c in CASES.items():
        with quiet():
            rc = RLmCfg(**copy.deepcopy(LLM))     # (the config class edits the rope_scaling dict it is given)
            rc.attn_implementation = "eager"
            lm = rlm.InternLM2ForCausalLM(rc).to(torch.bfloat16).eval()      # a fresh module per case: its rotary cache is stateful
        pre = "language_model."
        lm.load_state_dict({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}, strict=True)
        ids = prompt_ids(c["b"], c["prompt"], SEED + len(name))
        am = torch.ones_like(ids)
        tokens, gaps = [], []
        with torch.no_grad(), quiet():
            o = lm(input_ids=ids, attention_mask=am, position_ids=(am.cumsum(-1) - 1), use_cache=True)
            past = o.past_key_values
            for _ in range(c["new"]):
                row = o.logits[:, -1, :].float()
                top = row.topk(2, dim=-1).values
                nxt = row.argmax(-1)
                tokens.append(nxt)
                gaps.append(top[:, 0] - top[:, 1])
                am = torch.cat([am, torch.ones(c["b"], 1, dtype=torch.long)], 1)
                mi = lm.prepare_inputs_for_generation(nxt[:, None], past_key_values=past, attention_mask=am, use_cache=True)
                o = lm(**mi)
                past = o.past_key_values
        rot = lm.model.layers[0].attention.rotary_emb
        out["cases"][name] = dict(b=c["b"], prompt=c["prompt"], new=c["new"], ids_seed=SEED + len(name), ids=ids.clone(), tokens=torch.stack(tokens, 1),
                                  top2_gap=torch.stack(gaps, 1), cached_len=int(rot.max_seq_len_cached), cos_last=rot.cos_cached[-1].float().clone(),
                                  sin_last=rot.sin_cached[-1].float().clone())
        print(name, "tokens", torch.stack(tokens, 1).tolist(), "min top-2 gap", float(torch.stack(gaps, 1).min()), "rotary cache", rot.max_seq_len_cached)
    dst = os.path.join(HERE, "ntk_decode.pt")
    torch.save(out, dst)
    print("wrote", dst)


if __name__ == "__main__":
    main()
