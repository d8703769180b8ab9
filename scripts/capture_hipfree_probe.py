"""Is a device-memory release INSIDE a (relaxed) stream capture enough to invalidate it on ROCm 7.2?  A junk model's native context is destroyed (hipFree of its workspaces) in the middle
of a torch.cuda.graph capture on another stream; then the capture is ended and a kernel launched.  Also: torch's own tensor frees inside the capture (cached: no hipFree) as a control.
    python scripts/capture_hipfree_probe.py ctx|tensor|cudagraph"""
import gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import aigv_assessor_amd as pkg
from aigv_assessor_amd import native, synth
from aigv_assessor_amd.modeling import InternVLChatModel

what = sys.argv[1]
cfg = pkg.tiny(image_size=224, vit_layers=1, llm_layers=2)
junk = InternVLChatModel(cfg, max_clips=2)
junk.load_state_dict(synth.make_state_dict(cfg, seed=1, rich=True))
junk.eval().cuda()
toks = synth.canonical_tokens(cfg, 1, 2, seed=1)
junk.img_context_token_id = toks["img_context_token_id"]
kw = dict(mos=None, pixel_values=synth.synthetic_frames(2, 224, seed=2).cuda(), input_ids=toks["input_ids"], attention_mask=toks["attention_mask"], image_flags=torch.ones(2, 1, dtype=torch.long),
          labels=toks["labels"], motion_feature=synth.synthetic_motion(1, cfg.motion_dim, seed=2).cuda())
junk(**kw); torch.cuda.synchronize()
x = torch.ones(1024, device="cuda")
other = None
if what == "cudagraph":          # a second, already captured graph object to destroy inside the capture
    other = torch.cuda.CUDAGraph()
    with torch.cuda.graph(other):
        y0 = x * 2
    torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, capture_error_mode="relaxed"):
        y = x + 1
        if what == "ctx":
            native.load().aigv_ctx_destroy(junk._ctx); junk._ctx = None
        elif what == "tensor":
            big = torch.empty(1 << 28, device="cuda"); del big
        elif what == "cudagraph":
            del other, y0; gc.collect()
        z = y * 2
    g.replay(); torch.cuda.synchronize()
    print(f"{what}: capture survived, replay ok ({z[0].item()})")
except Exception as e:
    print(f"{what}: capture FAILED - {type(e).__name__}: {str(e).splitlines()[0][:160]}")
try:
    print("   a launch afterwards:", (x + 1).sum().item())
except Exception as e:
    print(f"   a launch afterwards FAILED - {type(e).__name__}: {str(e).splitlines()[0][:120]}")
os._exit(0)
