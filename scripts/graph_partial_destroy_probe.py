"""Which destruction of a captured graph poisons a LATER capture on this stack (ROCm 7.2, torch 2.10)?  One scenario per process (a poisoned process cannot launch any more):

    python scripts/graph_partial_destroy_probe.py forward      one model, forward graphs of three call shapes; ONE graph destroyed; a fourth shape captured
    python scripts/graph_partial_destroy_probe.py lookahead    the same through the look-ahead loop (dp_front graph captured under the prefetch stream); the dp_front graph destroyed
    python scripts/graph_partial_destroy_probe.py two_models   two models with graphs; ALL graphs of model A dropped (A.set_gemm_mode); model B captures a new shape
    python scripts/graph_partial_destroy_probe.py del_model    two models with graphs; model A deleted; model B captures a new shape
Prints OK or the failure."""
import gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import aigv_assessor_amd as pkg
from aigv_assessor_amd import eval_utils, synth
from aigv_assessor_amd.modeling import InternVLChatModel
from aigv_assessor_amd.slowfast import SlowFastR50

mode = sys.argv[1]
big = "--8b" in sys.argv
cfg = pkg.internvl2_8b() if big else pkg.tiny(image_size=224, vit_layers=2, llm_layers=2)
T = 8


def make():
    m = InternVLChatModel(cfg, max_clips=2, device=torch.device("cuda", 0))
    m.load_state_dict(synth.make_state_dict(cfg, seed=1, rich=True, device="cuda"))
    m.eval()
    m.slowfast_model = SlowFastR50(synth.slowfast_state_dict(seed=3))
    m.enable_graph_replay(True)
    return m


def shape(i):
    toks = synth.canonical_tokens(cfg, 1, T, seed=1)
    ids, lab = toks["input_ids"], toks["labels"]
    a0 = int((lab[0] != -100).nonzero()[0])
    ids = torch.cat([ids[:, :a0], torch.full((1, i), 7), ids[:, a0:]], 1)
    lab = torch.cat([lab[:, :a0], torch.full((1, i), -100), lab[:, a0:]], 1)
    return ids, lab, toks["img_context_token_id"]


pv = synth.synthetic_frames(T, cfg.image_size, seed=2).cuda()
flags = torch.ones(T, 1, dtype=torch.long)


def fwd(m, i, n=3):
    ids, lab, ctx = shape(i)
    m.img_context_token_id = ctx
    out = None
    for _ in range(n):
        out = m(mos=None, pixel_values=pv, input_ids=ids, attention_mask=torch.ones_like(ids, dtype=torch.bool), image_flags=flags, labels=lab)
        torch.cuda.synchronize()
    return out["score1"].item()


def loop(m, i, n=12):
    ids, lab, ctx = shape(i)
    m.img_context_token_id = ctx
    items = [{"input_ids": ids, "labels": lab, "attention_mask": torch.ones_like(ids, dtype=torch.bool), "image_flags": torch.ones(1, T, 1, dtype=torch.long), "pixel_values": pv.float().cpu()[None]}
             for _ in range(n)]
    return [o["score1"].item() for _, o in eval_utils.batched(items, m, k=2)][-1]


try:
    a = make()
    if mode == "forward":
        ref = [fwd(a, i) for i in range(3)]
        key = next(k for k, v in a._graphs.items() if isinstance(v, tuple))
        torch.cuda.synchronize(); a._graphs.pop(key); gc.collect()
        new = fwd(a, 3); again = fwd(a, 1)
        assert again == ref[1]
    elif mode == "lookahead":
        ref = [loop(a, i) for i in range(3)]
        key = next(k for k, v in a._graphs.items() if isinstance(v, tuple) and k[0][0] == "dp_front")
        torch.cuda.synchronize(); a._graphs.pop(key); gc.collect()
        new = loop(a, 3); again = loop(a, 1)
        assert again == ref[1]
    elif mode in ("two_models", "del_model"):
        b = make()
        ra, rb = loop(a, 0), loop(b, 1)
        if mode == "two_models":
            a.set_gemm_mode(-1)
        else:
            del a; gc.collect(); torch.cuda.synchronize()
        new = loop(b, 2); again = loop(b, 1)
        assert again == rb
        if mode == "two_models":
            assert loop(a, 0) == ra
    print(f"{mode}: OK (captured graphs now: {sum(isinstance(v, tuple) for v in (b if mode in ('two_models', 'del_model') else a)._graphs.values())})", flush=True)
except Exception as e:
    print(f"{mode}: FAILED - {type(e).__name__}: {str(e).splitlines()[0][:200]}", flush=True)
os._exit(0)
