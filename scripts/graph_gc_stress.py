"""Root cause check for the now-and-then invalidated capture (round 6): Python's CYCLIC garbage collector running a native finalizer (aigv_ctx_destroy -> hipFree) INSIDE a stream capture.
Cycles of: a model with a native context becomes cyclic garbage (never collected explicitly), then another model captures a pass - with the collector's thresholds set low so that it
fires often.  `protected` = the library as it is (native.capturing(): collect first, collector off during the capture, releases parked); `unprotected` = that bracket replaced by a no-op.

    python scripts/graph_gc_stress.py protected|unprotected [cycles = 40]"""
import contextlib, gc, os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import aigv_assessor_amd as pkg
from aigv_assessor_amd import native, synth
from aigv_assessor_amd.modeling import InternVLChatModel

mode, cycles = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40
if mode == "unprotected":
    native.capturing = contextlib.nullcontext
    native.release = lambda fn, h: getattr(native.load(), fn)(h) if h is not None else None
cfg = pkg.tiny(image_size=224, vit_layers=1, llm_layers=2)
sd = synth.make_state_dict(cfg, seed=1, rich=True)
T = 2
toks = synth.canonical_tokens(cfg, 1, T, seed=1)
pv = synth.synthetic_frames(T, 224, seed=2).cuda()
mo = synth.synthetic_motion(1, cfg.motion_dim, seed=2).cuda()
flags = torch.ones(T, 1, dtype=torch.long)


def make():
    m = InternVLChatModel(cfg, max_clips=2)
    m.load_state_dict(sd)
    m.eval().cuda()
    m.img_context_token_id = toks["img_context_token_id"]
    return m


def run(m):
    o = m(mos=None, pixel_values=pv, input_ids=toks["input_ids"], attention_mask=toks["attention_mask"], image_flags=flags, labels=toks["labels"], motion_feature=mo)
    torch.cuda.synchronize()
    return o["score1"].item()


gc.set_threshold(50, 2, 2)          # the collector fires every few dozen allocations: if a finalizer can land inside a capture, it will
warnings.simplefilter("error")
ref = None
for c in range(cycles):
    try:
        junk = make(); run(junk)       # a model with a live native context ...
        del junk                       # ... now cyclic garbage (modules refer to their owner): only the cyclic collector frees it
        m = make()
        m.enable_graph_replay(True)
        vals = [run(m) for _ in range(3)]          # eager, CAPTURE, replay
        ref = ref or vals
        assert vals == ref and any(isinstance(v, tuple) for v in m._graphs.values())
        m.enable_graph_replay(False)
    except Exception as e:
        print(f"{mode}: cycle {c}: {type(e).__name__}: {str(e).splitlines()[0][:170]}", flush=True)
        os._exit(1)
print(f"{mode}: {cycles} cycles clean", flush=True)
os._exit(0)
