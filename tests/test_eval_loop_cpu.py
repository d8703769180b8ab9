"""Host logic of the batched / sharded eval loop (eval_utils.batched, shard, gather_rows) on CPU: a
stand-in model built from the CPU oracle takes the place of the HIP model (which cannot run here), so what is checked is the grouping, the
collator-style padding, the per-item slicing of the group's outputs, the one-group-ahead ordering and the rank sharding - the bit-equality
of the real kernels in and out of a batch is the GPU suite's business (tests/test_gpu_api.py)."""
import itertools
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import aigv_assessor_amd as pkg
from aigv_assessor_amd import eval_utils, synth
from oracle import oracle as O


class _Ahead:
    def __init__(self, pv, n_clips):
        self.pv, self.n_clips = pv, n_clips


class OracleLoopModel:
    """The members the loop helpers use: device, ingest_frames, prefetch, __call__ -> {'score1', 'logit', 'label'} as InternVLChatModel
    returns them (argmax ids on consumed rows, -1 elsewhere; the score row relative to each clip's un-padded end)."""

    def __init__(self, cfg, sd, ctx_id):
        self.cfg, self.sd, self.ctx = cfg, sd, ctx_id
        self.device = torch.device("cpu")
        self.calls = []            # (n_clips, padded width) of every forward
        self.order = []            # 'front' / 'score' events: the one-group-ahead ordering

    def ingest_frames(self, frames_u8):
        parts = list(frames_u8) if isinstance(frames_u8, (list, tuple)) else [frames_u8]
        return O.normalize_frames_u8(torch.cat(parts))

    def prefetch(self, pixel_values=None, frames_u8=None, n_clips=1):
        self.order.append("front")
        return _Ahead(self.ingest_frames(frames_u8) if frames_u8 is not None else pixel_values, n_clips)

    def __call__(self, mos=None, pixel_values=None, input_ids=None, attention_mask=None, image_flags=None, labels=None, motion_feature=None):
        self.order.append("score")
        pv = pixel_values.pv if isinstance(pixel_values, _Ahead) else pixel_values
        B, N = input_ids.shape
        self.calls.append((B, N))
        scores, logits = [], []
        T = pv.shape[0] // B
        for b in range(B):        # every clip on its own un-padded row: what "a clip's bits do not depend on its batch mates" means
            m = attention_mask[b].bool()
            n = int(m.sum())
            ids, lab = input_ids[b][m][None], labels[b][m][None]
            ref = O.forward_eval(self.sd, self.cfg, pv[b * T:(b + 1) * T].float(), ids, torch.ones_like(ids, dtype=torch.bool),
                                 torch.ones(T, 1, dtype=torch.long), lab, motion_feature[b:b + 1], self.ctx, stage=2)
            row = torch.full((N - 1,), -1, dtype=torch.long)
            pos = m.nonzero().flatten()[:-1]
            keep = ref["label"] != -100
            row[pos[keep]] = ref["logit"][keep]
            logits.append(row)
            scores.append(ref["score1"].reshape(1))
        return {"score1": torch.cat(scores).to(torch.bfloat16), "logit": torch.cat(logits), "label": labels[:, 1:].reshape(-1)}


def _rig(n_items=7, T=2, ragged_geometry=False):
    cfg = pkg.tiny(vit_hidden=64, vit_heads=1, vit_layers=1, vit_inter=128, llm_hidden=256, llm_heads=2, llm_kv_heads=1,
                   llm_layers=1, llm_inter=256, vocab=256, image_size=56, score_dims=(32, 1), motion_dim=128)
    sd = synth.make_state_dict(cfg, seed=11, dtype=torch.float32, rich=True)
    items = []
    for i in range(n_items):
        toks = synth.canonical_tokens(cfg, 1, T, seed=20 + i)
        ids, lab, am = toks["input_ids"], toks["labels"], toks["attention_mask"]
        extra = i % 3                                             # ragged prompts: a few more question tokens in front of the answer
        if extra:
            a0 = int((lab[0] != -100).nonzero()[0])
            fill = torch.randint(3, cfg.llm_config.vocab_size - 8, (1, extra), generator=torch.Generator().manual_seed(i))
            ids = torch.cat([ids[:, :a0], fill, ids[:, a0:]], 1)
            lab = torch.cat([lab[:, :a0], torch.full((1, extra), -100), lab[:, a0:]], 1)
            am = torch.ones_like(ids, dtype=am.dtype)
        size = 56 if not (ragged_geometry and i in (3, 4)) else 40
        g = torch.Generator().manual_seed(100 + i)
        item = {"input_ids": ids, "labels": lab, "attention_mask": am, "image_flags": torch.ones(1, T, 1, dtype=torch.long),
                "mos": torch.tensor([0.1 * i]), "motion_feature": synth.synthetic_motion(1, 128, seed=30 + i, dtype=torch.float32),
                "video_name": [f"clip{i}.mp4"], "answer": ["good"]}
        if ragged_geometry:
            item["frames"] = torch.randint(0, 256, (T, size, size, 3), dtype=torch.uint8, generator=g)
        else:
            item["pixel_values"] = synth.synthetic_frames(T, 56, seed=100 + i, dtype=torch.float32)[None]
        items.append(item)
    return cfg, sd, items, synth.canonical_tokens(cfg, 1, T, seed=20)["img_context_token_id"]


def _plain_loop(model, items, frames=None):
    """The reference's loop shape (stage2_eval.py:908-941), one clip per call."""
    rows = []
    for it in items:
        pv = model.ingest_frames(frames(it)) if frames else it["pixel_values"][0].to(torch.bfloat16)      # (the bf16 cast of stage2_eval.py:932)
        out = model(mos=None, pixel_values=pv, input_ids=it["input_ids"], attention_mask=it["attention_mask"], image_flags=it["image_flags"][0],
                    labels=it["labels"], motion_feature=it["motion_feature"])
        rows.append((it["video_name"][0], out["score1"].clone(), out["logit"].clone(), out["label"].clone()))
    return rows


@pytest.mark.parametrize("k,ahead", [(1, False), (3, True), (4, True), (4, False), (16, True)])
def test_batched_loop_yields_what_the_plain_loop_yields(k, ahead):
    cfg, sd, items, ctx = _rig()
    want = _plain_loop(OracleLoopModel(cfg, sd, ctx), items)
    model = OracleLoopModel(cfg, sd, ctx)
    got = list(eval_utils.batched(items, model, k=k, ahead=ahead))
    assert [it["video_name"][0] for it, _ in got] == [w[0] for w in want]                   # every item, in order
    for (it, out), (_, score, logit, label) in zip(got, want):
        n = it["input_ids"].shape[1]
        assert out["logit"].shape == (n - 1,) and out["label"].shape == (n - 1,) and out["score1"].shape == (1,)
        assert torch.equal(out["score1"], score) and torch.equal(out["logit"], logit) and torch.equal(out["label"], label)
        assert torch.equal(eval_utils.answer_ids(it["labels"][0], out["logit"]), eval_utils.answer_ids(it["labels"][0], logit))
        assert out["loss"].item() == pytest.approx(abs(out["score1"].float().item() - it["mos"].to(torch.bfloat16).float().item()), abs=2e-2)
    groups = [min(k, len(items) - i) for i in range(0, len(items), k)]
    assert [b for b, _ in model.calls] == groups                                            # ONE forward per group of k
    assert all(n == max(it["input_ids"].shape[1] for it in items[i:i + k]) for (_, n), i in zip(model.calls, range(0, len(items), k)))
    if ahead and len(groups) > 1:      # group g + 1's visual front is enqueued BEFORE group g is scored
        assert model.order[:3] == ["front", "front", "score"] and model.order.count("front") == len(groups)
    if not ahead:
        assert "front" not in model.order


def test_batched_loop_over_a_real_dataloader_with_the_drivers_item_layout():
    """The driver's own plumbing (stage2_eval.py:906-941): a map-style dataset whose items are UNBATCHED tensors / strings / floats, wrapped by
    ``DataLoader(batch_size=1)`` (default collate: leading 1 on tensors, strings into lists, floats into a [1] tensor) - sharded, batched, and
    equal to the plain loop over the same loader."""
    from torch.utils.data import DataLoader, Dataset
    cfg, sd, items, ctx = _rig(n_items=6)

    class DS(Dataset):
        def __len__(self):
            return len(items)

        def __getitem__(self, i):
            it = items[i]
            return {"input_ids": it["input_ids"][0], "labels": it["labels"][0], "attention_mask": it["attention_mask"][0], "image_flags": it["image_flags"][0],
                    "pixel_values": it["pixel_values"][0], "motion_feature": it["motion_feature"][0], "mos": float(it["mos"]), "video_name": it["video_name"][0],
                    "answer": "The static quality of the video is good."}
    want = _plain_loop(OracleLoopModel(cfg, sd, ctx), list(DataLoader(DS(), batch_size=1)))
    for world in (1, 2):
        rows = []
        for rank in range(world):
            model = OracleLoopModel(cfg, sd, ctx)
            loader = DataLoader(eval_utils.shard(DS(), rank, world), batch_size=1)
            got = list(eval_utils.batched(loader, model, k=2))
            assert all(isinstance(it["video_name"], list) and it["input_ids"].dim() == 2 and it["mos"].shape == (1,) for it, _ in got)
            rows.append([(it["video_name"][0], out["score1"], out["logit"]) for it, out in got])
        merged = [rows[j % world][j // world] for j in range(len(items))]
        for (name, score, logit), (wname, wscore, wlogit, _) in zip(merged, want):
            assert name == wname and torch.equal(score, wscore) and torch.equal(logit, wlogit)


def test_score_dataset_is_the_drivers_loop_body():
    """eval_utils.score_dataset = stage2_eval.py:908-973 as one call: rows [video_name, answer, output, mos, score1, level] equal to the plain loop's,
    CSV written with the driver's columns, the six correlation figures + substring accuracy returned."""
    import csv
    import tempfile

    class Tok:          # token id -> one of the five level words (ids mod 5), joined by blanks: enough for the slice / decode / parse chain
        words = ("bad", "poor", "fair", "good", "excellent")

        def decode(self, ids):
            return " ".join(self.words[int(i) % 5] for i in ids.tolist() if int(i) >= 0)
    cfg, sd, items, ctx = _rig(n_items=5)
    for it, a in zip(items, ("bad", "poor", "fair", "good", "excellent")):
        it["answer"] = [f"The quality of the video is {a}."]
    im_end = synth.canonical_tokens(cfg, 1, 2, seed=20)["im_end_id"]
    want = []
    for it, (name, score, logit, _label) in zip(items, _plain_loop(OracleLoopModel(cfg, sd, ctx), items)):
        text = Tok().decode(eval_utils.answer_ids(it["labels"][0], logit, im_end_id=im_end))
        want.append([name, it["answer"][0], text, float(it["mos"]), float(score.float().item()), eval_utils.parse_level(text)])
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "out.csv")
        rows, metrics = eval_utils.score_dataset(items, OracleLoopModel(cfg, sd, ctx), Tok(), k=2, output_file=path, im_end_id=im_end)
        assert rows == want
        got = list(csv.reader(open(path)))
        assert tuple(got[0]) == eval_utils.CSV_COLUMNS and len(got) == 1 + len(items) and got[1][0] == "clip0.mp4"
    assert set(metrics) == {"acc", "level_srcc", "level_plcc", "level_krcc", "pred_score_srcc", "pred_score_plcc", "pred_score_krcc"}
    assert eval_utils.score_dataset([], None, Tok()) == ([], None)


def test_groups_break_where_the_frame_geometry_changes_and_uint8_frames_are_ingested():
    cfg, sd, items, ctx = _rig(n_items=7, ragged_geometry=True)          # items 3, 4 hold 40x40 frames, the others 56x56
    fr = lambda it: it["frames"]

    class M(OracleLoopModel):
        def ingest_frames(self, frames_u8):                               # (resize stand-in: the loop logic is what is tested)
            parts = list(frames_u8) if isinstance(frames_u8, (list, tuple)) else [frames_u8]
            x = torch.cat(parts).permute(0, 3, 1, 2).float()
            x = torch.nn.functional.interpolate(x, size=(56, 56), mode="nearest").permute(0, 2, 3, 1).to(torch.uint8)
            return O.normalize_frames_u8(x)
    want = _plain_loop(M(cfg, sd, ctx), items, frames=fr)
    model = M(cfg, sd, ctx)
    got = list(eval_utils.batched(items, model, k=4, frames=fr))
    assert [b for b, _ in model.calls] == [3, 2, 2]                       # 0-2 | 3-4 (another frame size) | 5-6
    for (it, out), (_, score, logit, _) in zip(got, want):
        assert torch.equal(out["score1"], score) and torch.equal(out["logit"], logit)


def test_shard_partitions_the_set_and_empty_input_yields_nothing():
    assert list(eval_utils.batched([], None)) == []
    data = list(range(11))
    for world in (1, 2, 3, 8, 16):
        parts = [list(eval_utils.shard(data, r, world)) for r in range(world)]
        assert sorted(itertools.chain(*parts)) == data
        assert all(p == data[r::world] for r, p in enumerate(parts))
        gen = [list(eval_utils.shard(iter(data), r, world)) for r in range(world)]         # a plain iterable: strided iterator
        assert gen == parts
    from torch.utils.data import DataLoader, Dataset

    class DS(Dataset):
        touched = []

        def __len__(self):
            return 5

        def __getitem__(self, i):
            DS.touched.append(i)
            return {"x": torch.tensor([i])}
    dl = DataLoader(eval_utils.shard(DS(), 1, 2), batch_size=1)
    assert [int(b["x"]) for b in dl] == [1, 3] and DS.touched == [1, 3]                      # the other rank's items are never loaded
    with pytest.raises(ValueError):
        eval_utils.shard(data, 2, 2)
    with pytest.raises(ValueError):
        next(eval_utils.batched(data, None, k=0))


def _shard_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    dist.init_process_group("gloo")
    cfg, sd, items, ctx = _rig(n_items=7)
    model = OracleLoopModel(cfg, sd, ctx)
    rows = []
    for it, out in eval_utils.batched(eval_utils.shard(items, rank, world), model, k=2):
        rows.append((it["video_name"][0], "good", "good", float(it["mos"]), out["score1"].float().item(), 4))
    allrows = eval_utils.gather_rows(rows)
    q.put((rank, len(rows), allrows, sum(b for b, _ in model.calls)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_batched_loop_over_gloo_scores_every_clip_once(world):
    """N ranks, each scoring ITS share in groups of 2, then ONE all_gather_object of the result rows: every rank ends with the rows of the
    whole set, in the set's order, equal to the one-process plain loop's - and no clip was scored twice (stage2_eval.py:908-911 scores the
    whole set on every rank)."""
    cfg, sd, items, ctx = _rig(n_items=7)
    want = _plain_loop(OracleLoopModel(cfg, sd, ctx), items)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_shard_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sum(n for _, n, _, _ in res) == len(items) and sum(c for _, _, _, c in res) == len(items)
    for rank, _, allrows, _ in res:
        assert [r[0] for r in allrows] == [w[0] for w in want]
        assert [r[4] for r in allrows] == [w[1].float().item() for w in want]
    stats = eval_utils.save_and_evaluate(res[0][2])
    assert set(stats) >= {"acc", "pred_score_srcc", "pred_score_plcc"}


def test_shard_and_interleave_are_inverse_for_every_set_size_and_world():
    """eval_utils.shard / gather_rows' interleave for every (set size 0..40, world 1..9): the shares are disjoint, cover the set, differ in size by at most one, and
    interleaving the per-rank rows gives the set back in its order - for map-style datasets (Subset) and plain iterables (islice) alike."""
    from aigv_assessor_amd import eval_utils

    class DS:                                   # map-style: __len__ + __getitem__
        def __init__(self, n):
            self.n = n

        def __len__(self):
            return self.n

        def __getitem__(self, i):
            if not 0 <= i < self.n:
                raise IndexError(i)
            return i
    for world in range(1, 10):
        for n in range(0, 41):
            for make in (lambda: DS(n), lambda: iter(range(n))):
                parts = [[x for x in eval_utils.shard(make(), r, world)] for r in range(world)]
                assert sorted(x for p in parts for x in p) == list(range(n))
                assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
                assert eval_utils._interleave(parts) == list(range(n)), (n, world)
    with pytest.raises(ValueError):
        eval_utils.shard(range(4), 3, 3)
