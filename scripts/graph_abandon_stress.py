"""Does a capture ABANDONED by a Python exception (valid capture, ended by torch, graph thrown away) make a LATER capture fail now and then?  Cycles of: an abandoned capture through
InternVLChatModel._graph_call, then a fresh call shape captured and replayed.   python scripts/graph_abandon_stress.py [cycles = 100]"""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import aigv_assessor_amd as pkg
from aigv_assessor_amd import synth
from aigv_assessor_amd.modeling import InternVLChatModel

cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 100
cfg = pkg.tiny(image_size=224, vit_layers=1, llm_layers=2)
model = InternVLChatModel(cfg, max_clips=2)
model.load_state_dict(synth.make_state_dict(cfg, seed=1, rich=True))
model.eval().cuda()
InternVLChatModel.GRAPH_CACHE_SIZE = 4 * cycles + 8
model.enable_graph_replay(True)
T = 2
base = synth.canonical_tokens(cfg, 1, T, seed=1)
model.img_context_token_id = base["img_context_token_id"]
pv = synth.synthetic_frames(T, 224, seed=2).cuda()
mo = synth.synthetic_motion(1, cfg.motion_dim, seed=2).cuda()
x = torch.ones(4, device="cuda")
for c in range(cycles):
    def flaky(t):
        if torch.cuda.is_current_stream_capturing():
            raise ValueError("no")
        return t + 1
    model._graph_call(("flaky", c), [x], flaky)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model._graph_call(("flaky", c), [x], flaky)           # the abandoned capture
    ids, lab = base["input_ids"], base["labels"]
    a0 = int((lab[0] != -100).nonzero()[0])
    ids = torch.cat([ids[:, :a0], torch.full((1, c % 40), 7), ids[:, a0:]], 1)
    lab = torch.cat([lab[:, :a0], torch.full((1, c % 40), -100), lab[:, a0:]], 1)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            for _ in range(3):
                o = model(mos=None, pixel_values=pv, input_ids=ids, attention_mask=torch.ones_like(ids, dtype=torch.bool), image_flags=torch.ones(T, 1, dtype=torch.long), labels=lab, motion_feature=mo)
                torch.cuda.synchronize()
            if c >= 40:
                model._drop_graphs()                               # (40 prompt lengths: start over, parked)
    except Exception as e:
        print(f"cycle {c}: {type(e).__name__}: {str(e).splitlines()[0][:160]}", flush=True)
        os._exit(1)
print(f"{cycles} cycles clean", flush=True)
os._exit(0)
