"""Child process of tests/test_gpu_dist.py (not a test module): first contact of the data-parallel scorer with RCCL on ONE MI355X.

A fresh process (the process group and the GPU are initialised here, never in the pytest parent) with WORLD_SIZE = 1:
``init_dist('pytorch', 'nccl')`` (internvl/dist_utils.py:32-104 in the reference; backend 'nccl' is RCCL on ROCm), then
``score_clips_dp`` with the collectives FORCED although one rank needs none - ``all_gather_into_tensor`` of the pre-projector visual
tokens with an async work handle, the result gathers - against the plain ``forward`` of the same model.  Prints RCCL_OK on success."""
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    import torch
    import torch.distributed as dist
    import aigv_assessor_amd as pkg
    from aigv_assessor_amd import dist_utils, synth
    from aigv_assessor_amd.modeling import InternVLChatModel

    dist_utils.init_dist("pytorch", backend="nccl")
    assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
    dev = torch.device("cuda", torch.cuda.current_device())
    # 1. the collective helper on device tensors: async handle, completion on the compute stream
    dist_utils.force_single_rank_collectives = True
    x = torch.arange(6 * 5 * 8, dtype=torch.float32, device=dev).reshape(6, 5, 8).to(torch.bfloat16)
    finish = dist_utils.all_gather_rows_begin(x, [6])
    y = (x.float() * 2).to(torch.bfloat16)            # work enqueued between begin and finish overlaps with the collective
    g = finish().all()
    assert g.data_ptr() != x.data_ptr() and torch.equal(g, x) and torch.equal(y.float(), x.float() * 2)
    # 2. the scorer through it
    cfg = pkg.tiny(image_size=224, vit_layers=1, llm_layers=2)
    sd = synth.make_state_dict(cfg, seed=71, rich=True)
    model = InternVLChatModel(cfg)
    model.load_state_dict(sd)
    model.eval().cuda()
    B, T = 3, 2
    toks = synth.canonical_tokens(cfg, B, T, seed=71)
    model.img_context_token_id = toks["img_context_token_id"]
    pv = synth.synthetic_frames(B * T, 224, seed=71).to(dev)
    motion = synth.synthetic_motion(B, cfg.motion_dim, seed=71).to(dev)
    flags = torch.ones(B * T, 1, dtype=torch.long)
    plain = model(mos=None, pixel_values=pv, input_ids=toks["input_ids"], attention_mask=toks["attention_mask"], image_flags=flags,
                  labels=toks["labels"], motion_feature=motion)
    dp = dist_utils.score_clips_dp(model, pv, toks["input_ids"], toks["attention_mask"], flags, toks["labels"], motion)
    # the same with the LLM pass fed from the buffer the RCCL all-gather produced (not from the rank's own shard)
    dpg = dist_utils.score_clips_dp(model, pv, toks["input_ids"], toks["attention_mask"], flags, toks["labels"], motion, prefer_gathered=True)
    torch.cuda.synchronize()
    assert torch.equal(dp["score1"], plain["score1"]) and torch.equal(dp["logit"], plain["logit"]) and torch.equal(dp["label"], plain["label"])
    assert torch.equal(dpg["score1"], plain["score1"]) and torch.equal(dpg["logit"], plain["logit"])
    # 2b. the same with HIP-graph replay: the front half (ViT shard [+ SlowFast]) and the projector + InternLM2 half are each ONE captured graph,
    #     the RCCL all-gather runs between them on the host's side (calls 1-2 eager / capture, then replays) - still the plain forward's bits
    model.enable_graph_replay(True)
    for i in range(4):
        d2 = dist_utils.score_clips_dp(model, pv, toks["input_ids"], toks["attention_mask"], flags, toks["labels"], motion, prefer_gathered=bool(i & 1))
        torch.cuda.synchronize()
        assert torch.equal(d2["score1"], plain["score1"]) and torch.equal(d2["logit"], plain["logit"]), i
    assert sum(1 for v in model._graphs.values() if isinstance(v, tuple)) >= 2, list(model._graphs.values())
    model.enable_graph_replay(False)
    # 3. a rank-0-style reduction the driver uses (bench.py: all_reduce(MAX) of the step time)
    t = torch.tensor([1.5], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t.item()) == 1.5
    dist.barrier()
    dist.destroy_process_group()
    print(f"RCCL_OK backend=nccl nccl_version={torch.cuda.nccl.version()} device={torch.cuda.get_device_name(dev)}")


if __name__ == "__main__":
    main()
