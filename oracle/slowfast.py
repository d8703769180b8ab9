"""CPU ORACLE for the SlowFast-R50 motion branch  —  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.

**PARITY UNPINNED.**  The reference builds this branch from a third-party package that is absent from
/root/reference and from this image: ``pytorchvideo.models.hub.slowfast_r50(pretrained=True)`` (CHAT:23, CHAT:164; version
not pinned - the package is not even listed in requirements.txt), and its weights are a network download.  What follows
restates the PUBLISHED architecture (Feichtenhofer et al., "SlowFast Networks for Video Recognition", ICCV 2019, the
R50 8x8 instantiation, as built by pytorchvideo's ``create_slowfast`` defaults) and is anchored on the reference's own
call sites:

    CHAT = internvl/model/internvl_chat_eval2/modeling_internvl_chat.py
      CHAT:97-133   pack_pathway_output: fast = all T frames, slow = frames[linspace(0, T-1, T//4).long()]
      CHAT:164-177  feature_extraction = blocks 0..4 of the hub model; slow/fast AvgPool3d of block 5; AdaptiveAvgPool3d of block 6
      CHAT:179-193  forward: blocks -> repeat_interleave(4, dim=2) on both pathways -> AvgPool3d((8,7,7)) / ((32,7,7)), stride 1
                    -> AdaptiveAvgPool3d(1) -> cat -> [B, 2304, 1, 1, 1]

It cannot be checked against the real package here; the only external anchor is the parameter count of the published
model (34.57 M including the 2304x400 classifier; tests/test_oracle_golden.py checks 34.57 M - 0.92 M for these blocks).

Architecture (channels slow/fast; BatchNorm3d eps 1e-5 in eval mode after every conv; no conv bias):
  block 0   stem: conv [1,7,7] -> 64 / [5,7,7] -> 8, stride [1,2,2], pad [k//2]; ReLU; MaxPool [1,3,3] stride [1,2,2] pad [0,1,1];
            then fuse: fast -> conv [7,1,1] stride [4,1,1] pad [3,0,0] -> 2x channels, BN, ReLU, concatenated AFTER the slow channels
  blocks 1-4  res2..res5 with (3,4,6,3) bottleneck blocks: conv_a [kt,1,1] (kt slow = 1,1,3,3; fast = 3), BN, ReLU;
            conv_b [1,3,3] (spatial stride 1,2,2,2 in the first block of the stage), BN, ReLU; conv_c 1x1x1, BN;
            shortcut 1x1x1 conv + BN with the same stride when the shape changes; ReLU(shortcut + branch);
            inner widths 64/8 doubling per stage, out = 4 x inner; the fuse of block 0 repeats after res2, res3, res4.
State-dict names are pytorchvideo's module names under the reference's attribute path
``slowfast_model.feature_extraction.<block>.`` (CHAT:171-173).
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
DEPTHS = (3, 4, 6, 3)
PREFIX = "slowfast_model.feature_extraction."


def _bn(sd: Dict[str, Tensor], p: str, x: Tensor) -> Tensor:
    dt = x.dtype
    return F.batch_norm(x, sd[p + ".running_mean"].to(dt), sd[p + ".running_var"].to(dt), sd[p + ".weight"].to(dt),
                        sd[p + ".bias"].to(dt), False, 0.0, 1e-5)


def _conv(sd: Dict[str, Tensor], p: str, x: Tensor, stride, pad) -> Tensor:
    return F.conv3d(x, sd[p + ".weight"].to(x.dtype), None, stride, pad)


def _fuse(sd, p: str, xs: Tensor, xf: Tensor) -> Tensor:
    """FuseFastToSlow: time-strided conv of the fast pathway, concatenated after the slow channels."""
    f = F.relu(_bn(sd, p + ".norm", _conv(sd, p + ".conv_fast_to_slow", xf, (4, 1, 1), (3, 0, 0))))
    return torch.cat([xs, f], dim=1)


def _res_block(sd, p: str, x: Tensor, s: int) -> Tensor:
    sc = x
    if p + ".branch1_conv.weight" in sd:
        sc = _bn(sd, p + ".branch1_norm", _conv(sd, p + ".branch1_conv", x, (1, s, s), 0))
    kt = sd[p + ".branch2.conv_a.weight"].shape[2]
    y = F.relu(_bn(sd, p + ".branch2.norm_a", _conv(sd, p + ".branch2.conv_a", x, 1, (kt // 2, 0, 0))))
    y = F.relu(_bn(sd, p + ".branch2.norm_b", _conv(sd, p + ".branch2.conv_b", y, (1, s, s), (0, 1, 1))))
    y = _bn(sd, p + ".branch2.norm_c", _conv(sd, p + ".branch2.conv_c", y, 1, 0))
    return F.relu(sc + y)


def pack_pathways(frames: Tensor):
    """CHAT:97-133.  frames [B, 3, T, H, W] -> [slow, fast]."""
    T = frames.shape[2]
    idx = torch.linspace(0, T - 1, T // 4).long()
    return [frames.index_select(2, idx), frames]


def slowfast_blocks(sd: Dict[str, Tensor], frames: Tensor, prefix: str = PREFIX):
    """Blocks 0..4 (CHAT:171-173, 182).  Returns the slow [B,2048,T/4,h,w] and fast [B,256,T,h,w] maps, in frames.dtype."""
    xs, xf = pack_pathways(frames)
    x = [xs, xf]
    for i in (0, 1):
        p = f"{prefix}0.multipathway_blocks.{i}"
        kt = sd[p + ".conv.weight"].shape[2]
        y = F.relu(_bn(sd, p + ".norm", _conv(sd, p + ".conv", x[i], (1, 2, 2), (kt // 2, 3, 3))))
        x[i] = F.max_pool3d(y, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    x[0] = _fuse(sd, f"{prefix}0.multipathway_fusion", x[0], x[1])
    for stage in range(4):
        for i in (0, 1):
            for blk in range(DEPTHS[stage]):
                s = 2 if (blk == 0 and stage > 0) else 1
                x[i] = _res_block(sd, f"{prefix}{stage + 1}.multipathway_blocks.{i}.res_blocks.{blk}", x[i], s)
        if stage < 3:
            x[0] = _fuse(sd, f"{prefix}{stage + 1}.multipathway_fusion", x[0], x[1])
    return x


def slowfast_features(sd: Dict[str, Tensor], frames: Tensor, prefix: str = PREFIX) -> Tensor:
    """CHAT:179-193: frames [B, 3, T, H, W] (the model dtype; bf16 in the eval driver) -> motion feature [B, 2304]."""
    xs, xf = slowfast_blocks(sd, frames, prefix)
    xs = xs.repeat_interleave(4, dim=2)
    xf = xf.repeat_interleave(4, dim=2)

    def pools(x: Tensor, kt: int) -> Tensor:
        # torch CPU has no bf16 avg_pool3d (the reference's bf16 model only runs this on a GPU): pool in fp32 and round to the
        # model dtype after each of the two modules, which is what a bf16 kernel with fp32 accumulation returns
        dt = x.dtype
        y = F.avg_pool3d(x.float(), (kt, 7, 7), (1, 1, 1)).to(dt)
        return F.adaptive_avg_pool3d(y.float(), 1).to(dt)

    return torch.cat([pools(xs, 8), pools(xf, 32)], dim=1).flatten(1)
