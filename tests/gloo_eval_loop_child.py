"""Child process of tests/test_gpu_dist.py (not a test module): rank RANK of WORLD_SIZE ranks on the ONE MI355X, running the reference's
eval loop the way INTEGRATION.md shows it for N GPUs - ``eval_utils.shard`` gives the rank its share of the set, ``eval_utils.batched`` scores
it in groups of k with the real model, ``eval_utils.gather_rows`` (one all_gather_object over gloo; RCCL refuses duplicate devices) puts every
rank's result rows back into the set's order - against the one-process plain loop (one clip per call) run by the same rank.  No data-path
collective: each clip is scored on one rank by the kernels that score it alone.  Prints LOOP_OK on success."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    import torch
    import torch.distributed as dist
    import aigv_assessor_amd as pkg
    from aigv_assessor_amd import dist_utils, eval_utils, synth
    from aigv_assessor_amd.modeling import InternVLChatModel
    from aigv_assessor_amd.slowfast import SlowFastR50

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ["LOCAL_RANK"] = "0"
    dist_utils.init_dist("pytorch", backend="gloo")
    cfg = pkg.tiny(image_size=224, vit_layers=2, llm_layers=2)
    model = InternVLChatModel(cfg, max_clips=3)
    model.load_state_dict(synth.make_state_dict(cfg, seed=97, rich=True))
    model.eval().cuda()
    model.slowfast_model = SlowFastR50(synth.slowfast_state_dict(seed=3))
    T, n_items = 8, 7
    g = torch.Generator().manual_seed(97)
    items = []
    for i in range(n_items):                                  # (every rank builds the same set: a dataset)
        toks = synth.canonical_tokens(cfg, 1, T, seed=97 + i)
        items.append({"input_ids": toks["input_ids"], "labels": toks["labels"], "attention_mask": toks["attention_mask"],
                      "image_flags": torch.ones(1, T, 1, dtype=torch.long), "mos": torch.tensor([0.1 * i]), "video_name": [f"clip{i}"],
                      "frames": torch.randint(0, 256, (T, 240, 320, 3), dtype=torch.uint8, generator=g).pin_memory()})
    model.img_context_token_id = toks["img_context_token_id"]
    plain = []
    for it in items:
        o = model(mos=None, pixel_values=model.ingest_frames(it["frames"].cuda()), input_ids=it["input_ids"], attention_mask=it["attention_mask"],
                  image_flags=it["image_flags"][0], labels=it["labels"])
        plain.append((it["video_name"][0], o["score1"].float().item(), eval_utils.answer_ids(it["labels"][0], o["logit"].cpu()).tolist()))
    rows = [(it["video_name"][0], o["score1"].float().item(), eval_utils.answer_ids(it["labels"][0], o["logit"]).tolist())
            for it, o in eval_utils.batched(eval_utils.shard(items, rank, world), model, k=3, frames=lambda it: it["frames"])]
    assert len(rows) == len(range(rank, n_items, world))
    every = eval_utils.gather_rows(rows)
    assert every == plain, (rank, every, plain)
    dist.barrier()
    dist.destroy_process_group()
    print(f"LOOP_OK rank={rank}/{world} scored {len(rows)} of {n_items} clips")


if __name__ == "__main__":
    main()
