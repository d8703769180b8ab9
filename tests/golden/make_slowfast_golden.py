"""Regression fixture for oracle/slowfast.py (NOT a reference vector: pytorchvideo is absent offline, so this pins the restatement
against accidental change only - see the PARITY UNPINNED note in oracle/slowfast.py).

    python tests/golden/make_slowfast_golden.py      ->  tests/golden/slowfast_oracle.pt
"""
import os, sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import aigv_assessor_amd  # noqa: E402,F401
from aigv_assessor_amd import synth  # noqa: E402
from oracle import slowfast as OSF  # noqa: E402

torch.manual_seed(0)
sd = synth.slowfast_state_dict(seed=11)
frames = synth.synthetic_frames(12, 224, seed=11).view(1, 12, 3, 224, 224).permute(0, 2, 1, 3, 4).contiguous()
with torch.no_grad():
    f32 = OSF.slowfast_features(sd, frames.float())
    bf16 = OSF.slowfast_features(sd, frames)
xs, xf = OSF.slowfast_blocks(sd, frames.float())
torch.save({"seed": 11, "frames": 12, "size": 224, "feature_fp32": f32, "feature_bf16": bf16,
            "slow_map_mean": xs.mean(dim=(2, 3, 4)), "fast_map_mean": xf.mean(dim=(2, 3, 4))},
           os.path.join(ROOT, "tests", "golden", "slowfast_oracle.pt"))
print("feature mean", float(f32.mean()), "bf16 vs fp32 max", float((bf16.float() - f32).abs().max()))
