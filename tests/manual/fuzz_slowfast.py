"""Randomised run of the native SlowFast-R50 branch over the geometries it accepts (8 <= T <= 32 with T % 4 == 0; H, W multiples of 32 in 224..1024; 1-3 clips; non-square frames)
through the suite's own case (tests/test_gpu_slowfast.py::test_slowfast_branch_matches_oracle: against the fp32 and the bf16-module evaluations of oracle/slowfast.py, the reference's
[slow, fast] call form, one clip alone == in the batch).  (test infrastructure: uses oracle/; the CPU oracle takes seconds per case.)

    python tests/manual/fuzz_slowfast.py [n_cases = 16] [seed = 0]"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from aigv_assessor_amd import synth  # noqa: E402
from aigv_assessor_amd.slowfast import SlowFastR50  # noqa: E402
from oracle import slowfast as osf  # noqa: E402

BF = torch.bfloat16
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 16
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = random.Random(seed0)
sd = synth.slowfast_state_dict(seed=3)
sf = SlowFastR50(sd)
bad = 0
for c in range(n_cases):
    B, T = rng.randint(1, 3), rng.choice([8, 8, 12, 16, 20, 24, 28, 32])
    H, W = 32 * rng.randint(7, 14), 32 * rng.randint(7, 14)
    if rng.random() < 0.4:
        W = H
    while B * T * H * W > 3 * 16 * 320 * 320:               # keep the CPU oracle within seconds
        B = max(1, B - 1)
        if B == 1:
            T = min(T, 16)
            H, W = min(H, 352), min(W, 352)
    g = torch.Generator().manual_seed(seed0 * 1000 + c)
    frames = torch.randn(B * T, 3, H, W, generator=g).clamp(-2.5, 2.5).to(BF)
    clip = frames.view(B, T, 3, H, W).permute(0, 2, 1, 3, 4)
    want32 = osf.slowfast_features(sd, clip.float())
    want16 = osf.slowfast_features(sd, clip).float()
    got = sf.features(frames.cuda(), B).float().cpu()
    ref_err, err, scale = (want16 - want32).abs(), (got - want32).abs(), want32.abs().mean()
    ok = got.shape == (B, 2304) and bool(torch.isfinite(got).all())
    ok = ok and float(err.mean()) <= 1.5 * float(ref_err.mean()) + 1e-3 * float(scale) and float(err.max()) <= 2.0 * float(ref_err.max()) + 5e-3 * float(scale)
    fast = clip.cuda()
    again = sf([fast.index_select(2, torch.linspace(0, T - 1, T // 4).long().cuda()), fast])
    ok = ok and again.shape == (B, 2304, 1, 1, 1) and torch.equal(again.view(B, -1).float().cpu(), got)
    one = sf.features(frames[:T].cuda(), 1).float().cpu()
    ok = ok and torch.equal(one[0], got[0])
    print(f"case {c}: {B} clips x {T} x {H} x {W}: mean err {float(err.mean()):.5f} (bf16 modules {float(ref_err.mean()):.5f}), max {float(err.max()):.4f} ({float(ref_err.max()):.4f}) {'ok' if ok else 'FAILED'}", flush=True)
    bad += not ok
assert bad == 0, bad
print(f"FUZZ_SLOWFAST_OK {n_cases} cases; native handles cached {len(sf._handles)}, retired {sf.epoch}")
