#!/usr/bin/env python
"""Close the SlowFast-R50 parity loop on a machine that HAS pytorchvideo (this repo's build image does not: the branch is
"parity unpinned" in DESIGN.md §2 until this script has been run somewhere).

The reference builds its motion branch from ``pytorchvideo.models.hub.slowfast_r50(pretrained=True)``
(internvl/model/internvl_chat_eval2/modeling_internvl_chat.py:135-193).  This script takes that very module, runs one clip through

  (a) the reference's own data flow on the pytorchvideo modules   (pack_pathway_output :97-133, blocks 0..4, repeat_interleave(4),
      AvgPool3d((8|32,7,7)), AdaptiveAvgPool3d(1), concat :179-193 - restated here from those lines, 20 lines),
  (b) this repo's restatement of the architecture, ``oracle/slowfast.py``, fed pytorchvideo's state dict,
  (c) optionally (--gpu, on an MI355X) the native HIP branch ``aigv_assessor_amd.slowfast.SlowFastR50``,

and reports, block by block, the largest and mean relative difference of the slow / fast feature maps and of the final
[B, 2304] feature.  Expected: (a) vs (b) equal to fp32 round-off (same ops, same order); (c) within bf16 noise (1e-2 relative).

    pip install pytorchvideo          # (+ its torch / fvcore dependencies)
    python tools/check_slowfast_against_pytorchvideo.py [--pretrained] [--frames 8] [--size 448] [--gpu] [--save weights.pth]

``--save`` writes the blocks' state dict under the reference's names (``slowfast_model.feature_extraction.*``) - the file
``InternVLChatModel.load_state_dict`` / ``SlowFastR50(torch.load(...))`` accept.
Nothing in the package, the tests or bench.py imports this file.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def reference_flow(blocks, frames):
    """modeling_internvl_chat.py:97-133,179-193 on pytorchvideo's own modules; returns per-block (slow, fast) maps and the feature."""
    T = frames.shape[2]
    idx = torch.linspace(0, T - 1, T // 4).long()
    x = [frames.index_select(2, idx), frames]
    maps = []
    for i in range(5):
        x = blocks[i](x)
        maps.append((x[0].clone(), x[1].clone()))
    slow, fast = x[0].repeat_interleave(4, dim=2), x[1].repeat_interleave(4, dim=2)
    slow = torch.nn.AdaptiveAvgPool3d(1)(torch.nn.AvgPool3d((8, 7, 7), stride=1)(slow))
    fast = torch.nn.AdaptiveAvgPool3d(1)(torch.nn.AvgPool3d((32, 7, 7), stride=1)(fast))
    return maps, torch.cat([slow, fast], dim=1).flatten(1)


def rel(a, b):
    d = (a.float() - b.float()).abs()
    s = b.float().abs().mean().clamp_min(1e-12)
    return float(d.max() / s), float(d.mean() / s)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pretrained", action="store_true", help="download the hub weights (the reference does); default: the module's random init")
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--size", type=int, default=448)
    ap.add_argument("--gpu", action="store_true", help="also run the native HIP branch (needs an MI355X and the built library)")
    ap.add_argument("--save", default=None)
    args = ap.parse_args()
    try:
        from pytorchvideo.models.hub import slowfast_r50
    except ImportError as e:
        raise SystemExit(f"pytorchvideo is not installed here ({e}); this script exists for a machine that has it")
    from oracle import slowfast as OSF      # checker
    model = slowfast_r50(pretrained=args.pretrained).eval()
    blocks = model.blocks
    for m in model.modules():               # give BatchNorm non-trivial statistics when the weights are the random init
        if isinstance(m, torch.nn.BatchNorm3d) and not args.pretrained:
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    sd = {OSF.PREFIX + k[len("blocks."):]: v.detach().float() for k, v in model.state_dict().items()
          if k.startswith("blocks.") and int(k.split(".")[1]) <= 4}
    if args.save:
        torch.save(sd, args.save)
        print("wrote", args.save, f"({sum(v.numel() for v in sd.values() if v.is_floating_point()) / 1e6:.2f} M values)")
    g = torch.Generator().manual_seed(0)
    frames = torch.randn(1, 3, args.frames, args.size, args.size, generator=g).clamp_(-2.5, 2.5)
    with torch.no_grad():
        ref_maps, ref_feat = reference_flow(blocks, frames)
        xs, xf = OSF.slowfast_blocks(sd, frames, return_all=True) if "return_all" in OSF.slowfast_blocks.__code__.co_varnames else (None, None)
        feat = OSF.slowfast_features(sd, frames)
    if xs is not None:
        for i, ((rs, rf), (os_, of)) in enumerate(zip(ref_maps, zip(xs, xf))):
            print(f"block {i}: slow max/mean rel diff {rel(os_, rs)}, fast {rel(of, rf)}")
    else:
        os_, of = OSF.slowfast_blocks(sd, frames)
        print(f"block 4 (res5 output): slow max/mean rel diff {rel(os_, ref_maps[4][0])}, fast {rel(of, ref_maps[4][1])}")
    mx, mn = rel(feat, ref_feat)
    print(f"feature [1, 2304]: oracle/slowfast.py vs pytorchvideo max rel {mx:.3e}, mean rel {mn:.3e}  ->", "MATCH" if mx < 1e-4 else "MISMATCH")
    if args.gpu:
        from aigv_assessor_amd.slowfast import SlowFastR50
        pv = frames[0].permute(1, 0, 2, 3).contiguous().to(torch.bfloat16).cuda()       # [T, 3, S, S] like pixel_values
        got = SlowFastR50(sd).features(pv, 1).float().cpu()
        mx, mn = rel(got, ref_feat)
        print(f"feature [1, 2304]: native HIP branch vs pytorchvideo max rel {mx:.3e}, mean rel {mn:.3e}  ->", "MATCH (bf16 noise)" if mn < 2e-2 else "MISMATCH")


if __name__ == "__main__":
    main()
