"""Does destroying captured graphs in front of a new capture invalidate that capture now and then (ROCm 7.2 / torch 2.10)?  Cycles of: build a model, capture two passes (a forward
graph and, through the look-ahead loop, a dp_front graph), delete the model, build the next one, capture again.  `park` (the library's behaviour since round 6) keeps the dropped graph
objects alive; `destroy` lets them die with their model (the behaviour before).  Reports the cycle at which a capture was invalidated, if any - the process cannot go on after that.

    python scripts/graph_destroy_stress.py park|destroy [cycles = 80]"""
import gc, os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import aigv_assessor_amd as pkg
from aigv_assessor_amd import eval_utils, modeling, synth
from aigv_assessor_amd.modeling import InternVLChatModel
from aigv_assessor_amd.slowfast import SlowFastR50

mode, cycles = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 80
if mode == "destroy":
    class _NoPark(list):
        def extend(self, it):
            for _ in it:
                pass
    modeling._PARKED_GRAPHS = _NoPark()
cfg = pkg.tiny(image_size=224, vit_layers=2, llm_layers=2)
T = 8
sd = synth.make_state_dict(cfg, seed=1, rich=True)
sf = synth.slowfast_state_dict(seed=3)
toks = synth.canonical_tokens(cfg, 1, T, seed=1)
pv = synth.synthetic_frames(T, 224, seed=2)
items = [{"input_ids": toks["input_ids"], "labels": toks["labels"], "attention_mask": toks["attention_mask"], "image_flags": torch.ones(1, T, 1, dtype=torch.long), "pixel_values": pv.float()[None]}
         for _ in range(8)]
ref = None
warnings.simplefilter("error")          # a failed capture warns: make it fatal here
for c in range(cycles):
    try:
        m = InternVLChatModel(cfg, max_clips=2)
        m.load_state_dict(sd)
        m.eval().cuda()
        m.slowfast_model = SlowFastR50(sf)
        m.img_context_token_id = toks["img_context_token_id"]
        m.enable_graph_replay(True)
        vals = [o["score1"].item() for _, o in eval_utils.batched(items, m, k=2)]
        assert sum(isinstance(v, tuple) for v in m._graphs.values()) == 2, m._graphs.values()
        ref = ref or vals
        assert vals == ref
        del m
        gc.collect()
    except Exception as e:
        print(f"{mode}: cycle {c}: {type(e).__name__}: {str(e).splitlines()[0][:160]}", flush=True)
        os._exit(1)
print(f"{mode}: {cycles} cycles clean (parked graph objects: {len(modeling._PARKED_GRAPHS)})", flush=True)
os._exit(0)
