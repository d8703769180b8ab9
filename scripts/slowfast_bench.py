"""SlowFast-R50 motion branch timing at the headline shape (4 clips x 8 frames x 448 px): python scripts/slowfast_bench.py [clips]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aigv_assessor_amd
from aigv_assessor_amd import synth
from aigv_assessor_amd.slowfast import SlowFastR50
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
T, S = 8, 448
sf = SlowFastR50(synth.slowfast_state_dict(0))
x = synth.synthetic_frames(B * T, S, seed=1).cuda()
for _ in range(3): f = sf.features(x, B)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 20
e0.record()
for _ in range(n): f = sf.features(x, B)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
fl = sf.flops_per_clip() * B
print(f"slowfast B={B} T={T} {S}px: {ms:.3f} ms / batch, {fl/1e9:.1f} GFLOP -> {fl/ms/1e9:.1f} TFLOP/s; feature mean {f.float().mean().item():.4f}")
