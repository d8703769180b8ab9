"""Child process of tests/test_gpu_dist.py (not a test module): rank RANK of WORLD_SIZE ranks that all drive the SAME MI355X.

RCCL refuses two ranks on one device ("Duplicate GPU detected"), and a builder's box has one card, so the multi-rank DATA PATH of
score_clips_dp is rehearsed on hardware over gloo (the token shards cross through host memory: dist_utils' rehearsal transport): frames
split over the ranks independent of clip boundaries, every rank's real InternViT pass on its shard, the all-gather, clips split over the
ranks, the real projector + InternLM2 pass, the result gather - against the one-process forward of the same model.  What this does NOT
cover is RCCL itself at N > 1 (tests/rccl_single_rank_child.py covers RCCL at N = 1).  Prints DP_OK on success."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    import torch
    import torch.distributed as dist
    import aigv_assessor_amd as pkg
    from aigv_assessor_amd import dist_utils, synth
    from aigv_assessor_amd.modeling import InternVLChatModel

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ["LOCAL_RANK"] = "0"                      # every rank on device 0
    dist_utils.init_dist("pytorch", backend="gloo")
    assert dist.get_world_size() == world and dist.get_backend() == "gloo"
    dev = torch.device("cuda", 0)
    cfg = pkg.tiny(image_size=224, vit_layers=1, llm_layers=2)
    sd = synth.make_state_dict(cfg, seed=73, rich=True)
    model = InternVLChatModel(cfg)
    model.load_state_dict(sd)
    model.eval().cuda()
    # ragged clip / frame splits; one clip over all ranks (latency mode); even split; one 8-frame clip (3 + 3 + 2 frames on three ranks: the
    # ragged latency split of SURVEY 8e)
    for B, T in ((3, 2), (1, 4), (4, 2), (1, 8)):
        toks = synth.canonical_tokens(cfg, B, T, seed=73 + B)
        model.img_context_token_id = toks["img_context_token_id"]
        pv = synth.synthetic_frames(B * T, 224, seed=73 + B).to(dev)
        motion = synth.synthetic_motion(B, cfg.motion_dim, seed=73 + B).to(dev)
        flags = torch.ones(B * T, 1, dtype=torch.long)
        plain = model(mos=None, pixel_values=pv, input_ids=toks["input_ids"], attention_mask=toks["attention_mask"], image_flags=flags,
                      labels=toks["labels"], motion_feature=motion)
        for prefer in (False, True):
            dp = dist_utils.score_clips_dp(model, pv, toks["input_ids"], toks["attention_mask"], flags, toks["labels"], motion, prefer_gathered=prefer)
            torch.cuda.synchronize()
            assert torch.equal(dp["score1"], plain["score1"]), (rank, B, T, prefer, dp["score1"], plain["score1"])
            assert torch.equal(dp["logit"], plain["logit"]) and torch.equal(dp["label"], plain["label"]), (rank, B, T, prefer)
        # the same passes with HIP-graph replay (front half and projector + LLM half as captured graphs around the all-gather): ranks with
        # and without clips, ragged frame shards - every call still the one-process forward, bit for bit
        model.enable_graph_replay(True)
        for prefer in (False, True):                    # (prefer_gathered: the projector + LLM graph is fed from the all-gathered buffer on every rank)
            for i in range(4):
                dp = dist_utils.score_clips_dp(model, pv, toks["input_ids"], toks["attention_mask"], flags, toks["labels"], motion, prefer_gathered=prefer)
                torch.cuda.synchronize()
                assert torch.equal(dp["score1"], plain["score1"]) and torch.equal(dp["logit"], plain["logit"]), (rank, B, T, "graph", prefer, i)
        model.enable_graph_replay(False)
    dist.barrier()
    dist.destroy_process_group()
    print(f"DP_OK rank={rank}/{world} backend=gloo device={torch.cuda.get_device_name(dev)}")


if __name__ == "__main__":
    main()
