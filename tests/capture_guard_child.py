"""Child process of tests/test_gpu_api.py (not a test module; a child because a regression here leaves a process that can launch nothing any more): a native finalizer that fires
INSIDE a stream capture must be parked, not run.  A model with a live native context is dropped and collected in the middle of a captured pass of another model
(InternVLChatModel._graph_call under native.capturing()): the capture must survive, replay with the right result, and the parked context must be destroyed afterwards.
Prints GUARD_OK."""
import gc
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import aigv_assessor_amd as pkg
    from aigv_assessor_amd import native, synth
    from aigv_assessor_amd.modeling import InternVLChatModel
    cfg = pkg.tiny(image_size=224, vit_layers=1, llm_layers=2)
    sd = synth.make_state_dict(cfg, seed=1, rich=True)
    toks = synth.canonical_tokens(cfg, 1, 2, seed=1)
    kw = dict(mos=None, pixel_values=synth.synthetic_frames(2, 224, seed=2).cuda(), input_ids=toks["input_ids"], attention_mask=toks["attention_mask"],
              image_flags=torch.ones(2, 1, dtype=torch.long), labels=toks["labels"], motion_feature=synth.synthetic_motion(1, cfg.motion_dim, seed=2).cuda())

    def make():
        m = InternVLChatModel(cfg, max_clips=2)
        m.load_state_dict(sd)
        m.eval().cuda()
        m.img_context_token_id = toks["img_context_token_id"]
        return m
    model, junk = make(), make()
    want = model(**kw)["score1"].clone()
    junk(**kw)                                   # junk holds a live native context now
    torch.cuda.synchronize()
    holder = [junk]
    del junk
    model.enable_graph_replay(True)
    x = torch.ones(8, device="cuda")
    seen = {}

    def fn(t):
        if torch.cuda.is_current_stream_capturing():
            seen["parked_before"] = len(native._deferred_releases)
            holder.clear()
            gc.collect()                         # the junk model's finalizer runs HERE, inside the capture
            seen["parked_inside"] = len(native._deferred_releases)
        return t + 1
    assert model._graph_call(("guard",), [x], fn) is None                     # first occurrence: eager by contract
    out = model._graph_call(("guard",), [x], fn)                              # captured (+ replayed once)
    torch.cuda.synchronize()
    assert out is not None and torch.equal(out, x + 1), out
    assert seen == {"parked_before": 0, "parked_inside": 1}, seen             # the release was parked ...
    assert native._deferred_releases == [] and native._captures_underway == 0  # ... and carried out when the capture had ended
    out = model._graph_call(("guard",), [x * 3], fn)                          # a replay
    assert torch.equal(out, x * 3 + 1)
    got = [model(**kw)["score1"].clone() for _ in range(3)]                   # and the model still captures and replays its own passes
    torch.cuda.synchronize()
    assert all(torch.equal(g, want) for g in got) and any(isinstance(v, tuple) and k[0][0] == "forward" for k, v in model._graphs.items())
    print("GUARD_OK")


if __name__ == "__main__":
    main()
