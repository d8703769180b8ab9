"""CPU-only tests: host logic, the C-ABI surface, and agreement of the package's host-side weight preparation
with the oracle.  No compute calls into the library (there is no GPU here)."""
import ctypes
import os
import re
import subprocess

import pytest
import torch

import aigv_assessor_amd as pkg
from aigv_assessor_amd import native, synth
from aigv_assessor_amd.conversation import get_conv_template
from aigv_assessor_amd.modeling import InternVLChatModel, resized_pos_table, rope_tables
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    from aigv_assessor_amd import build
    return build.build()


def test_library_exports_every_declared_symbol(built):
    header = open(os.path.join(ROOT, "include", "aigv_amd.h")).read()
    declared = set(re.findall(r"\b(aigv_[a-z0-9_]+)\s*\(", header))
    declared -= {"aigv_ctx", "aigv_config"}
    assert declared == set(native.PROTOTYPES), declared ^ set(native.PROTOTYPES)
    lib = ctypes.CDLL(built)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.aigv_abi_version() == native.ABI_VERSION
    assert lib.aigv_sizeof_config() == ctypes.sizeof(native.AigvConfig)


def test_no_cpu_fallback_and_loud_errors(built):
    lib = ctypes.CDLL(built)
    lib.aigv_last_error.restype = ctypes.c_char_p
    cfg = native.AigvConfig()
    h = ctypes.c_void_p()
    if not torch.cuda.is_available():
        rc = lib.aigv_ctx_create(0, ctypes.byref(cfg), ctypes.byref(h))
        assert rc != 0 and b"no HIP device" in lib.aigv_last_error(None)
        model = InternVLChatModel(pkg.tiny(image_size=224))
        with pytest.raises(native.NativeError):
            model.extract_feature(torch.zeros(1, 3, 224, 224))


def test_documented_default_of_attention_numerics_is_the_librarys(built):
    """A boundary doc that contradicts the header is a boundary bug (INTEGRATION.md once said "1 (default)" while code and
    header had moved to 0).  aigv_get_attention_numerics(NULL) = the mode a fresh context starts in, read without a GPU; header, INTEGRATION.md
    and the Python mirror must all name that value as the default."""
    import inspect
    lib = native.load()
    d = lib.aigv_get_attention_numerics(None)
    assert d in (0, 1)
    header = open(os.path.join(ROOT, "include", "aigv_amd.h")).read()
    assert f"#define AIGV_ATTENTION_NUMERICS_DEFAULT {d}\n" in header
    words = {0: "the score matrix stays fp32", 1: "it carries"}
    assert re.search(rf"{d} \(default\) = {words[d]}", header), "the header's description of aigv_set_attention_numerics names another default"
    assert not re.search(rf"{1 - d} \(default", header[header.index("Numerics of the prefill attention"):header.index("int aigv_get_attention_numerics")])
    row = next(ln for ln in open(os.path.join(ROOT, "INTEGRATION.md")) if ln.startswith("| `aigv_set_attention_numerics`"))
    assert re.search(rf"\(new\) {d} \(default", row) and not re.search(rf"\b{1 - d} \(default", row), row
    py_default = inspect.signature(InternVLChatModel.set_attention_numerics).parameters["mode"].default
    assert {"fp32": 0, "reference": 1}[py_default] == d
    src = inspect.getsource(InternVLChatModel._upload)
    assert f'getattr(self, "_attn_numerics", {d})' in src       # what a freshly uploaded context is set to


def test_hashed_weight_set_is_pinned_and_chunk_independent():
    """synth.hashed_uniform / make_state_dict(method="hash") (round 6: the weight set of the oracle-only 26B fixture): the algorithm is pinned by
    value - a fixture recorded with it must keep meaning the same weights -, does not depend on the chunking, gives every tensor its own values
    (no tensor is an index permutation of another) and the documented statistics (uniform, std 0.02)."""
    a = synth.hashed_uniform((8,), 1, 0.02, dtype=torch.float32)
    want = [-0.019403910264372826, 0.03301652520895004, -0.007542225066572428, -0.0029996458906680346, -0.029941143468022346, 0.02123815193772316,
            0.017400067299604416, -0.02617489919066429]
    assert a.tolist() == want
    b = synth.hashed_uniform((3, 5), 123456789012, 0.02)
    assert b.float().flatten().tolist()[:5] == [-0.017578125, 0.0194091796875, -0.02685546875, 0.032958984375, 0.0281982421875]
    big = synth.hashed_uniform((700, 3000), 77, 0.02)
    assert torch.equal(big, synth.hashed_uniform((700, 3000), 77, 0.02, chunk=4099))
    assert abs(big.float().std().item() - 0.02) < 2e-4 and abs(big.float().mean().item()) < 1e-4 and big.float().abs().max().item() <= 0.02 * 12 ** 0.5
    other = synth.hashed_uniform((700, 3000), 77 + 7919, 0.02)
    assert not torch.equal(big.flatten().sort().values, other.flatten().sort().values)       # not a permutation of one another
    assert abs(torch.corrcoef(torch.stack([big.float().flatten(), other.float().flatten()]))[0, 1].item()) < 5e-3
    cfg = pkg.tiny(image_size=224)
    sd = synth.make_state_dict(cfg, seed=3, rich=True, method="hash")
    assert list(sd) == [n for n, _s, _k in synth.weight_shapes(cfg)]
    small_only = dict(synth.make_state_dict_iter(cfg, seed=3, rich=True, method="hash", big=False))
    assert set(small_only) == {n for n, _s, k in synth.weight_shapes(cfg) if k not in ("linear", "embed")} and all(torch.equal(small_only[k], sd[k]) for k in small_only)
    with pytest.raises(ValueError):
        synth.make_state_dict(cfg, method="bogus")


def test_gemm_row_band_planner(built):
    """Host logic of the GEMM dispatch (no GPU): the bands cover every row exactly once, split factors divide the K-tile
    count, forced modes are honoured, and the headline LLM shapes get whole rounds on the 256 kernel."""
    lib = native.load()

    def plan(M, N, K, epi):
        p, est = (ctypes.c_int * 7)(), ctypes.c_double()
        native.check(lib.aigv_plan_gemm(M, N, K, epi, p, ctypes.byref(est)))
        return list(p)[:6], est.value, p[6]

    shapes = [(8708, 28672, 4096, 4), (8708, 4096, 14336, 3), (8708, 4096, 4096, 3), (8708, 6144, 4096, 0), (32800, 3072, 1024, 0),
              (32800, 4096, 1024, 1), (32800, 1024, 4096, 2), (2177, 4096, 14336, 3), (1, 128, 64, 0), (300, 256, 128, 1), (515, 384, 1024, 2),
              (4354, 6144, 6144, 0), (8192, 4096, 4096, 5), (70, 512, 192, 4)]
    for M, N, K, epi in shapes:
        (top, mid, ms, last, kind, ls), est, right = plan(M, N, K, epi)
        assert est > 0 and right == 0
        if top < 0:
            assert N % 256 == 0 and (mid, last, kind) == (0, 0, 0)
            continue
        assert top * 256 + mid * 256 + last == M, (M, N, K, top, mid, last)
        assert (kind == 0) == (last == 0)
        if top or mid:
            assert N % 256 == 0
        if mid:
            assert ms >= 2 and (K // 64) % ms == 0
        if kind == 1:
            assert last <= 64 and ls == 1 and epi != 5
        if ls > 1:
            assert kind == 2 and (K // 64) % ls == 0 and epi != 5
    # 4 clips x 2177 tokens: the 256 kernel runs whole rounds of 256 tiles (w2: 32 row tiles x 16 = 2 rounds, the next 2 row tiles
    # as one round of 8 K slices), never the 2.125-round launch
    (top, mid, ms, last, kind, ls), _, _ = plan(8708, 4096, 14336, 3)
    assert (top * 16) % 256 == 0 and mid * 16 * ms <= 256 and top + mid == 34
    # InternViT-6B widths (N = 3200, 9600 = 256 j + 128): the first 256 j columns keep the 256 kernel, 128 columns go to the 128 kernel
    for N in (3200, 9600):
        (top, mid, ms, last, kind, ls), _, right = plan(16400, N, 3200, 0)
        assert right == 128 and (top > 0 or top == -1)
    assert plan(16400, 12800, 3200, 1)[2] == 0 and plan(100, 384, 128, 0)[2] in (0, 128)
    try:
        native.check(lib.aigv_tune_gemm(1, 0.0))
        assert plan(8708, 4096, 4096, 3)[0] == [0, 0, 0, 8708, 2, 1] and plan(16400, 3200, 3200, 0)[2] == 0
        native.check(lib.aigv_tune_gemm(2, 0.0))
        assert plan(8708, 4096, 4096, 3)[0][0] == -1
    finally:
        native.check(lib.aigv_tune_gemm(0, 0.0))
    assert lib.aigv_plan_gemm(100, 100, 64, 0, (ctypes.c_int * 7)(), None) != 0      # N % 128


def test_shared_prefix_planning_is_host_logic():
    """forward_shared_prefix's bookkeeping (no GPU): the shared prefix ends where the prompts start to differ, never inside the
    video tokens, and every consumed row (answer rows, the score row at -4) stays in the continuation."""
    cfg = pkg.tiny(image_size=224)
    model = InternVLChatModel(cfg)
    base = synth.canonical_tokens(cfg, 2, 2, seed=3)
    model.img_context_token_id = base["img_context_token_id"]
    prompts = synth.perspective_prompts(base, 3, seed=3, question_lens=(16, 9, 23))
    flags = torch.ones(4, 1, dtype=torch.long)
    plans = [model._plan(p["input_ids"], p["attention_mask"], p["labels"], flags, 4) for p in prompts]
    pre = model._shared_prefix_lengths(plans, 2)
    n0 = base["input_ids"].shape[1]
    cut = n0 - 16 - 10                                     # first question token of the canonical layout
    assert pre == [cut, cut]
    for pl in plans:
        for b in range(2):
            assert pl["last_ctx"][b] < pre[b]              # all <IMG_CONTEXT> tokens inside the prefix
            rows = [r - pl["cu"][b] for r in pl["logit_rows"] if pl["cu"][b] <= r < pl["cu"][b + 1]] + [pl["score_rows"][b] - pl["cu"][b]]
            assert min(rows) >= pre[b]
    # identical prompts: the prefix stops in front of the first consumed row, not at the end
    same = [plans[0], plans[0]]
    p_same = model._shared_prefix_lengths(same, 2)
    assert p_same[0] == min(min(r for r in plans[0]["logit_rows"] if r < plans[0]["cu"][1]), plans[0]["score_rows"][0])
    # a difference in front of the video tokens leaves nothing to share
    ids = prompts[1]["input_ids"].clone()
    ids[0, 3] = 5 if int(ids[0, 3]) != 5 else 6
    bad = model._plan(ids, prompts[1]["attention_mask"], prompts[1]["labels"], flags, 4)
    with pytest.raises(ValueError, match="diverge"):
        model._shared_prefix_lengths([plans[0], bad], 2)


def test_dead_trailing_tokens_are_planned_away():
    """Host bookkeeping (no GPU): tokens behind a clip's last consumed row are dropped from the packed batch, consumed rows keep
    pointing at the same tokens, visual slots are never dropped, and the switch restores the reference's full row set."""
    cfg = pkg.tiny(image_size=224)
    model = InternVLChatModel(cfg)
    t = synth.canonical_tokens(cfg, 2, 2, seed=5)
    model.img_context_token_id = t["img_context_token_id"]
    flags = torch.ones(4, 1, dtype=torch.long)
    n = t["input_ids"].shape[1]
    full = model._plan(t["input_ids"], t["attention_mask"], t["labels"], flags, 4, drop_dead_tail=False)
    trim = model._plan(t["input_ids"], t["attention_mask"], t["labels"], flags, 4)
    assert full["lens"] == [n, n] and trim["lens"] == [n - 1, n - 1]            # only the closing <|im_end|> is dead
    assert len(trim["logit_rows"]) == len(full["logit_rows"]) == 20
    for rt, rf in zip(trim["logit_rows"] + trim["score_rows"], full["logit_rows"] + full["score_rows"]):
        assert int(trim["ids_packed"][rt]) == int(full["ids_packed"][rf])       # same token behind every consumed row
    assert torch.equal(trim["slot"][trim["slot"] >= 0], full["slot"][full["slot"] >= 0])
    assert trim["last_ctx"] == full["last_ctx"]
    # nothing consumed behind the prompt (stage-2 score only): rows after the score row (-4) go
    nolab = model._plan(t["input_ids"], t["attention_mask"], torch.full_like(t["labels"], -100), flags, 4)
    assert nolab["lens"] == [n - 3, n - 3] and nolab["score_rows"] == [n - 4, 2 * (n - 3) - 1]
    # full_logits keeps every row; the model-level switch too
    assert model._plan(t["input_ids"], t["attention_mask"], t["labels"], flags, 4, full_logits=True)["lens"] == [n, n]
    model.drop_dead_tail = False
    assert model._plan(t["input_ids"], t["attention_mask"], t["labels"], flags, 4)["lens"] == [n, n]
    model.drop_dead_tail = True
    # a visual slot behind the last consumed row protects the tail of that clip
    ids = t["input_ids"].clone()
    lab = torch.full_like(t["labels"], -100)
    m1 = InternVLChatModel(cfg, stage=1)
    m1.img_context_token_id = t["img_context_token_id"]
    lab[:, 10] = ids[:, 10]                                                       # one early answer row, image tokens follow
    early = m1._plan(ids, t["attention_mask"], lab, flags, 4)
    assert early["lens"] == [n, n]


def test_product_path_never_imports_the_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, "aigv-assessor_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(root, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
                assert "/root/reference" not in txt, f


def test_product_library_reads_no_environment_variable():
    """a stray variable must not be able to re-roll a summation order.  Every experiment knob of the kernels is a field of
    the context (aigv_ctx_tune) or a process default set by an explicit call (aigv_tune_*); the native sources hold no getenv."""
    csrc = os.path.join(ROOT, "aigv-assessor_amd", "csrc")
    for f in sorted(os.listdir(csrc)) + [os.path.join("..", "..", "include", "aigv_amd.h")]:
        assert "getenv" not in open(os.path.join(csrc, f)).read(), f


def test_canonical_layout_and_special_ids():
    c8 = pkg.internvl2_8b()
    assert synth.canonical_len(c8, 8) == 2177 and synth.canonical_len(c8, 16) == 4281
    assert synth.special_ids(92553) == dict(im_end=92542, im_start=92543, img=92544, img_end=92545, img_context=92546)
    cfg = pkg.tiny(image_size=224)
    t = synth.canonical_tokens(cfg, 2, 4, seed=0)
    assert t["input_ids"].shape == (2, synth.canonical_len(cfg, 4))
    assert int((t["input_ids"][0] == t["img_context_token_id"]).sum()) == 4 * cfg.num_image_token + 1
    assert int((t["labels"][0] != -100).sum()) == 10 and t["labels"][0, -1] == t["im_end_id"]
    assert c8.num_image_token == 256 and c8.proj_in == 4096


def test_state_dict_names_and_module_surface():
    cfg = pkg.tiny(image_size=224)
    sd = synth.make_state_dict(cfg, 3, rich=True)
    m = InternVLChatModel(cfg)
    assert m.load_state_dict(sd) == ([], [])
    assert set(m.state_dict()) == set(sd)
    # the attribute surface the reference drivers touch (stage2_eval.py:810-885)
    assert len(m.vision_model.encoder.layers) == cfg.vision_config.num_hidden_layers
    assert m.language_model.get_output_embeddings().weight.shape[0] == cfg.llm_config.vocab_size
    for name in ("vision_model", "language_model", "mlp1", "motion_mlp", "mlpscore"):
        for p in getattr(m, name).parameters():
            p.requires_grad = False
    m.language_model.resize_token_embeddings(cfg.llm_config.vocab_size + 9)
    assert m.language_model.get_input_embeddings().weight.shape[0] == cfg.llm_config.vocab_size
    assert m.num_image_token == 64 and m.template == "internlm2-chat" and m.img_context_token_id is None
    m.vision_model.resize_pos_embeddings(224, 448, 14)
    assert m.vision_model.embeddings.position_embedding.shape[1] == 1025
    with pytest.raises(RuntimeError):
        m.load_state_dict({"bogus": torch.zeros(1)})


def test_host_weight_prep_matches_oracle():
    cos, sin = rope_tables(128, 1e6, 300, 32768, {"type": "dynamic", "factor": 2.0})
    oc, os_ = O.rope_tables(128, 1e6, 300, torch.bfloat16, 32768, {"type": "dynamic", "factor": 2.0})
    assert torch.equal(cos, oc[:, :64]) and torch.equal(cos, oc[:, 64:]) and torch.equal(sin, os_[:, :64])
    # dynamic NTK is keyed on the longest SINGLE sequence (modeling_internlm2.py:218-243), never on the table's row count:
    # a 900-row table (packed capacity of a batch) for sequences of at most 300 tokens under max_pos 512 is NOT rescaled ...
    dyn = {"type": "dynamic", "factor": 2.0}
    cos, sin = rope_tables(128, 1e6, 900, 512, dyn, seq_len=None)
    oc, os_ = O.rope_tables(128, 1e6, 300, torch.bfloat16, 512, dyn)
    assert torch.equal(cos[:300], oc[:, :64]) and torch.equal(sin[:300], os_[:, :64])
    # ... and a 700-token sequence is, with the base of ITS length, whatever the table holds
    cos, sin = rope_tables(128, 1e6, 900, 512, dyn, seq_len=700)
    oc, os_ = O.rope_tables(128, 1e6, 700, torch.bfloat16, 512, dyn)
    assert torch.equal(cos[:700], oc[:, :64]) and torch.equal(sin[:700], os_[:, :64])
    assert not torch.equal(cos[:300], rope_tables(128, 1e6, 300, 512, dyn)[0])
    pos = torch.randn(1, 1025, 32, generator=torch.Generator().manual_seed(0)).to(torch.bfloat16)
    assert torch.equal(resized_pos_table(pos, 32, 16), O.vit_pos_embed(pos, 32, 16, 16))
    assert torch.equal(resized_pos_table(pos, 32, 32), pos)       # identity at the native grid


def test_prompt_template_internlm2_chat():
    t = get_conv_template("internlm2-chat")
    t.append_message(t.roles[0], "<image>\nHow would you rate the static quality of this video?")
    t.append_message(t.roles[1], None)
    p = t.get_prompt()
    assert p.startswith("<|im_start|>system\n") and p.endswith("<|im_end|><|im_start|>assistant\n")
    assert "<|im_end|><|im_start|>user\n<image>\nHow would you rate" in p and "<|im_end|>\n" not in p
    assert t.stop_token_ids == [2, 92543, 92542] and t.sep == "<|im_end|>"
    with pytest.raises(KeyError):
        get_conv_template("phi3-chat")


def test_pack_and_config_roundtrip():
    ids = torch.tensor([[5, 6, 0, 0], [7, 8, 9, 0]])
    am = torch.tensor([[1, 1, 0, 0], [1, 1, 1, 0]])
    packed, cu, row_of = InternVLChatModel._pack(ids, am)
    assert packed.tolist() == [5, 6, 7, 8, 9] and cu == [0, 2, 5]
    assert row_of.tolist() == [[0, 1, -1, -1], [2, 3, 4, -1]]
    c = pkg.InternVLChatConfig.from_dict(pkg.internvl2_8b().to_dict())
    assert c.llm_config.intermediate_size == 14336 and c.vision_config.num_hidden_layers == 24
    c26 = pkg.internvl2_26b()
    assert c26.vision_config.norm_type == "rms_norm" and c26.llm_config.hidden_size == 6144


def test_lora_merge_matches_peft_rule():
    from aigv_assessor_amd.weights import merge_lora_state_dict, strip_peft_names
    g = torch.Generator().manual_seed(0)
    w = torch.randn(48, 32, generator=g).to(torch.bfloat16)
    a, b = torch.randn(8, 32, generator=g).to(torch.bfloat16), torch.randn(48, 8, generator=g).to(torch.bfloat16)
    base = {"language_model.base_model.model.model.layers.0.attention.wo.base_layer.weight": w,
            "language_model.base_model.model.model.norm.weight": torch.ones(32)}
    lora = {"language_model.base_model.model.model.layers.0.attention.wo.lora_A.default.weight": a,
            "language_model.base_model.model.model.layers.0.attention.wo.lora_B.default.weight": b}
    out = merge_lora_state_dict(base, lora, alpha_over_r=2.0)
    assert set(out) == {"language_model.model.layers.0.attention.wo.weight", "language_model.model.norm.weight"}
    want = (w.float() + 2.0 * (b.float() @ a.float())).to(torch.bfloat16)
    assert torch.equal(out["language_model.model.layers.0.attention.wo.weight"], want)
    assert "x.weight" in strip_peft_names({"x.base_layer.weight": w})
    with pytest.raises(KeyError):
        merge_lora_state_dict(base, {k: v for k, v in lora.items() if "lora_A" in k})


def test_eval_utils_follow_the_reference_driver(tmp_path):
    from aigv_assessor_amd import eval_utils as E
    labels = torch.tensor([-100, -100, 5, 6, 7, 92542])
    logit = torch.tensor([1, 2, 3, 4, 5, 6])
    assert E.answer_ids(labels, logit).tolist() == [3, 4, 5] == O.answer_slice(labels, logit, 92542).tolist()
    for text in ("The static quality of the video is good.", "excellent", "bad poor", "n/a", "fairly poor"):
        assert E.parse_level(text) == O.parse_level(text)
    rows = [["a.mp4", "quality is good.", "good", 70.0, 0.71, 4], ["b.mp4", "quality is bad.", "poor", 20.0, 0.25, 2],
            ["c.mp4", "quality is fair.", "fair", 50.0, 0.45, 3]]
    m = E.save_and_evaluate(rows, str(tmp_path / "r.csv"))
    assert abs(m["acc"] - 2 / 3) < 1e-9 and m["level_srcc"] > 0.99 and m["pred_score_plcc"] > 0.9
    assert open(tmp_path / "r.csv").readline().strip() == "video_name,answer,output,mos,pred_score,level"


def test_oracle_frame_normalisation_definition():
    u = torch.randint(0, 256, (2, 4, 6, 3), generator=torch.Generator().manual_seed(1), dtype=torch.uint8)
    y = O.normalize_frames_u8(u)
    assert y.shape == (2, 3, 4, 6) and y.dtype == torch.bfloat16
    x = u[1, 2, 3, 1].float() / 255
    assert y[1, 1, 2, 3] == ((x - 0.456) / 0.224).to(torch.bfloat16)


def test_slowfast_host_mirror_state_dict_handling():
    """Host side of the motion branch (no GPU): prefix normalisation, dropped tensors, the freeze-loop surface, loud failure on CPU."""
    import pytest
    import torch
    from aigv_assessor_amd import synth
    from aigv_assessor_amd.slowfast import PREFIX, SlowFastR50
    sd = synth.slowfast_state_dict(seed=2)
    sf = SlowFastR50(sd)
    n_w = sum(1 for k in sd if not k.endswith("num_batches_tracked"))
    assert len(sf.state_dict()) == n_w and all(k.startswith(PREFIX) for k in sf.state_dict())
    # pytorchvideo's own names (blocks.N....) and the classifier / pool blocks 5, 6 that the reference drops
    pv = {"blocks." + k[len(PREFIX):]: v for k, v in sd.items()}
    pv["blocks.6.proj.weight"] = torch.zeros(400, 2304)
    pv["blocks.6.proj.bias"] = torch.zeros(400)
    sf2 = SlowFastR50(pv)
    assert sf2.state_dict().keys() == sf.state_dict().keys()
    assert all(torch.equal(sf2.state_dict()[k], v) for k, v in sf.state_dict().items())
    ps = list(sf.parameters())
    assert len(ps) == n_w and not any(p.requires_grad for p in ps)
    with pytest.raises(RuntimeError, match="no SlowFast tensors"):
        SlowFastR50({"vision_model.x": torch.zeros(1)})
    with pytest.raises(RuntimeError, match="GPU only"):
        sf.features(torch.zeros(8, 3, 224, 224), 1)


def test_model_state_dict_with_slowfast_tensors_builds_the_branch():
    import aigv_assessor_amd as pkg
    from aigv_assessor_amd import synth
    from aigv_assessor_amd.modeling import InternVLChatModel
    from aigv_assessor_amd.slowfast import SlowFastR50
    cfg = pkg.tiny(image_size=224, vit_layers=1, llm_layers=1)
    m = InternVLChatModel(cfg)
    assert m.slowfast_model is None
    m.load_state_dict({**synth.make_state_dict(cfg, seed=1), **synth.slowfast_state_dict(seed=1)})
    assert isinstance(m.slowfast_model, SlowFastR50)
    with pytest.raises(ValueError, match="precision"):
        m.set_precision("int4")
    m.set_precision("fp8")          # no context yet: remembered and applied when the context is built
    assert m._precision == "fp8"


def test_from_pretrained_reads_only_the_model_shards(tmp_path):
    """A checkpoint directory as the reference trainer leaves it (HF Trainer files + stage2_train.py:223-235's
    lora_weights.pth): only the model shards are loaded, the LoRA file is merged, the rest is ignored."""
    import json
    from safetensors.torch import save_file
    from aigv_assessor_amd.modeling import InternVLChatModel
    cfg = pkg.tiny(image_size=224, vit_layers=1, llm_layers=1)
    sd = synth.make_state_dict(cfg, seed=5, rich=True)
    keys = sorted(sd)
    half = len(keys) // 2
    d = tmp_path / "ckpt"
    d.mkdir()
    save_file({k: sd[k].contiguous() for k in keys[:half]}, str(d / "model-00001-of-00002.safetensors"))
    save_file({k: sd[k].contiguous() for k in keys[half:]}, str(d / "model-00002-of-00002.safetensors"))
    wm = {k: ("model-00001-of-00002.safetensors" if i < half else "model-00002-of-00002.safetensors") for i, k in enumerate(keys)}
    (d / "model.safetensors.index.json").write_text(json.dumps({"metadata": {}, "weight_map": wm}))
    (d / "config.json").write_text(json.dumps(cfg.to_dict()))
    # what a Trainer run leaves beside the shards: none of it may be read as weights
    torch.save({"not": "a tensor dict", "lr": 1e-4}, str(d / "training_args.bin"))
    torch.save({"state": {0: {"exp_avg": torch.zeros(3)}}}, str(d / "optimizer.pt"))
    torch.save({"last_epoch": 3}, str(d / "scheduler.pt"))
    torch.save({"cpu": torch.zeros(8, dtype=torch.uint8)}, str(d / "rng_state.pth"))
    assert InternVLChatModel._checkpoint_files(str(d)) == ["model-00001-of-00002.safetensors", "model-00002-of-00002.safetensors"]
    m = InternVLChatModel.from_pretrained(str(d), torch_dtype=torch.bfloat16)
    got = m.state_dict()
    assert set(got) == set(sd) and all(torch.equal(got[k], sd[k]) for k in sd)
    # + a LoRA adapter file: folded into the named weight by W + 2 * B @ A
    name = "language_model.model.layers.0.attention.wo"
    g = torch.Generator().manual_seed(1)
    a = torch.randn(8, sd[name + ".weight"].shape[1], generator=g) * 0.05
    b = torch.randn(sd[name + ".weight"].shape[0], 8, generator=g) * 0.05
    torch.save({"language_model.base_model.model.model.layers.0.attention.wo.lora_A.default.weight": a,
                "language_model.base_model.model.model.layers.0.attention.wo.lora_B.default.weight": b}, str(d / "lora_weights.pth"))
    m2 = InternVLChatModel.from_pretrained(str(d), torch_dtype=torch.bfloat16)
    want = (sd[name + ".weight"].float() + 2.0 * (b @ a)).to(torch.bfloat16)
    assert torch.equal(m2.state_dict()[name + ".weight"], want)
    assert torch.equal(m2.state_dict()["mlp1.1.weight"], sd["mlp1.1.weight"])
    # a pytorch_model.bin checkpoint, and a directory without shards
    d2 = tmp_path / "bin"
    d2.mkdir()
    torch.save(sd, str(d2 / "pytorch_model.bin"))
    torch.save({"x": 1}, str(d2 / "training_args.bin"))
    (d2 / "config.json").write_text(json.dumps(cfg.to_dict()))
    m3 = InternVLChatModel.from_pretrained(str(d2))
    assert all(torch.equal(m3.state_dict()[k], sd[k]) for k in sd)
    d3 = tmp_path / "empty"
    d3.mkdir()
    (d3 / "config.json").write_text(json.dumps(cfg.to_dict()))
    torch.save({"x": 1}, str(d3 / "training_args.bin"))
    with pytest.raises(FileNotFoundError):
        InternVLChatModel.from_pretrained(str(d3))


def test_build_inputs_and_get_index_match_the_reference(golden_dir):
    """prompts.build_inputs / get_index against outputs of the reference's own preprocess_internlm (dataset.py:595-682) and
    get_index (stage2_eval.py:429-441), recorded by tests/golden/make_host_golden.py with the stub tokenizer."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from stub_tokenizer import StubTokenizer
    from aigv_assessor_amd import prompts
    from aigv_assessor_amd.conversation import get_conv_template
    g = torch.load(os.path.join(golden_dir, "host_inputs.pt"), weights_only=True)
    tok = StubTokenizer(92553)
    assert len(g["samples"]) >= 6
    for s in g["samples"]:
        got = prompts.build_inputs(tok, s["question"], s["answer"], s["n_frames"], s["num_image_token"])
        assert torch.equal(got["input_ids"], s["input_ids"]), (s["n_frames"], s["question"])
        assert torch.equal(got["labels"], s["labels"])
        assert torch.equal(got["attention_mask"], s["attention_mask"])
        n_ctx = int((got["input_ids"] == tok.special["<IMG_CONTEXT>"]).sum())
        if s["n_frames"] * s["num_image_token"] + 1 < 4096 - 100:
            assert n_ctx == s["n_frames"] * s["num_image_token"] + 1            # 256 per frame + ONE motion slot (stage2_eval.py:493-494)
            lab = got["labels"][got["labels"] != -100]
            assert tok.decode(lab) == s["answer"].strip() + "<|im_end|>"           # answer + closing <|im_end|> (dataset.py:643-666)
        else:
            # 16 frames do not fit max_seq_length 4096: the reference truncates and ends with every label ignored
            assert got["input_ids"].numel() == 4096 and bool((got["labels"] == -100).all())
    assert len(g["get_index"]) >= 100
    for c in g["get_index"]:
        assert prompts.get_index(c["bound"], c["fps"], c["max_frame"], 0, c["num_segments"]) == c["indices"], c
    # the chat template, as chat() renders it
    t = get_conv_template("internlm2-chat")
    assert t.sep == g["chat_prompts"]["sep"] and t.system_message == g["chat_prompts"]["system_message"]
    t.append_message(t.roles[0], "<image>\nDescribe the quality.")
    t.append_message(t.roles[1], None)
    assert t.get_prompt() == g["chat_prompts"]["single"]
    t = get_conv_template("internlm2-chat")
    t.append_message(t.roles[0], "<image>\nDescribe the quality.")
    t.append_message(t.roles[1], "It is fair.")
    t.append_message(t.roles[0], "And the motion?")
    t.append_message(t.roles[1], None)
    assert t.get_prompt() == g["chat_prompts"]["history"]
    # batching pads like the reference's collator and the plan strips it again
    b = prompts.batch_inputs([prompts.build_inputs(tok, q, a, 4, 64) for q, a in (("Q one?", "Short."), ("A longer question?", "A longer answer."))])
    assert b["input_ids"].shape[0] == 2 and int(b["attention_mask"][0].sum()) < b["input_ids"].shape[1] == int(b["attention_mask"][1].sum())
    assert bool((b["labels"][0][~b["attention_mask"][0]] == -100).all()) and bool((b["input_ids"][0][~b["attention_mask"][0]] == 2).all())


@pytest.mark.parametrize("temperature,top_k,top_p", [(1.0, 50, 1.0), (0.7, 5, 0.9), (1.3, 0, 0.5), (1.0, 1, 1.0), (0.5, 1000, 0.05), (1.0, 0, 0.999)])
def test_sampling_warpers_match_transformers(temperature, top_k, top_p):
    """generate(do_sample=True) warps the logits like HF's multinomial sampling does (the reference hands its generation_config to HF,
    modeling_internvl_chat.py:798-809): temperature, top-k, top-p in HF's order - against transformers' own warper classes, on logits with
    ties and a dominant token."""
    from transformers.generation.logits_process import TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper
    from aigv_assessor_amd.modeling import InternVLChatModel
    g = torch.Generator().manual_seed(int(temperature * 10) + top_k)
    logits = torch.randn(5, 300, generator=g) * 3
    logits[1, :40] = logits[1, 0]            # ties across the top-k boundary
    logits[2, 7] = 40.0                      # one token holds nearly all the mass
    ids = torch.zeros(5, 1, dtype=torch.long)
    want = logits.clone()
    if temperature != 1.0:
        want = TemperatureLogitsWarper(temperature)(ids, want)
    if top_k > 0:
        want = TopKLogitsWarper(top_k=top_k, min_tokens_to_keep=1)(ids, want)
    if top_p < 1.0:
        want = TopPLogitsWarper(top_p=top_p, min_tokens_to_keep=1)(ids, want)
    got = InternVLChatModel._warp(logits.clone(), temperature, top_k, top_p)
    assert torch.equal(got, want)
    assert torch.isfinite(got).sum(-1).min() >= 1


def test_generation_logits_processors_match_transformers():
    """The host glue of generate(): HF's RepetitionPenaltyLogitsProcessor and NoRepeatNGramLogitsProcessor (the reference's generate()
    hands its generation_config to HF, modeling_internvl_chat.py:798-809).  The processors here are checked against transformers'
    own classes on random logits and token histories (a third-party dependency of the reference, installed here as 5.x; the
    reference pins 4.37.2 - the two classes' arithmetic has not changed)."""
    from transformers.generation.logits_process import NoRepeatNGramLogitsProcessor, RepetitionPenaltyLogitsProcessor
    g = torch.Generator().manual_seed(5)
    B, V = 3, 50
    for cur in (0, 1, 2, 5, 12):
        hist = torch.randint(0, 6, (B, cur), generator=g)              # few distinct tokens: repeated n-grams do occur
        logits = torch.randn(B, V, generator=g)
        for pen in (1.3, 0.7):
            ours = InternVLChatModel._repetition_penalty(pen)(hist, logits.clone())
            theirs = RepetitionPenaltyLogitsProcessor(pen)(hist, logits.clone()) if cur else logits
            assert torch.equal(ours, theirs)
        for n in (1, 2, 3):
            ours = InternVLChatModel._no_repeat_ngram(n)(hist, logits.clone())
            theirs = NoRepeatNGramLogitsProcessor(n)(hist, logits.clone())
            assert torch.equal(ours, theirs), (cur, n)
    mx, eos, pad, sampler, procs, beams = InternVLChatModel._gen_args(dict(max_new_tokens=7, eos_token_id=[5, 9], repetition_penalty=1.2, no_repeat_ngram_size=2), {})
    assert (mx, eos, pad, sampler, len(procs)) == (7, [5, 9], None, None, 2)
    assert InternVLChatModel._gen_args(None, dict(repetition_penalty=1.0, no_repeat_ngram_size=0))[4] == []
    assert beams is None
    assert InternVLChatModel._gen_args(dict(num_beams=3, length_penalty=0.5, early_stopping=True), {})[5] == dict(num_beams=3, length_penalty=0.5, early_stopping=True)
    assert InternVLChatModel._gen_args(dict(num_beams=2), {})[5] == dict(num_beams=2, length_penalty=1.0, early_stopping=False)
    with pytest.raises(NotImplementedError):
        InternVLChatModel._gen_args(dict(num_beams=2, do_sample=True), {})
    with pytest.raises(NotImplementedError):
        InternVLChatModel._gen_args(dict(num_beams=2, num_return_sequences=2), {})


def test_full_size_round3_fixture_is_plain_data(golden_dir):
    """tests/golden/e2e_8b_r3.pt (recorded from the imported reference by make_golden_8b_r3.py): loads with weights_only=True and holds
    the cases the -m gpu tests consume."""
    path = os.path.join(golden_dir, "e2e_8b_r3.pt")
    g = torch.load(path, weights_only=True)
    assert g["llm_config"]["num_hidden_layers"] == 32 and g["vision_config"]["num_hidden_layers"] == 24
    for key in ("batch4/fp32", "batch4/bf16", "stage1/fp32", "stage1/bf16", "greedy/bf16"):
        assert key in g["cases"], key
    b4 = g["cases"]["batch4/bf16"]
    assert b4["B"] == 4 and b4["T"] == 8 and b4["seed"] == 0 and b4["score1"].shape == (4,) and b4["logit"].numel() == 40
    assert b4["hidden_m4"].shape == (4, 4096) and b4["top_ids"].shape == (40, 4)
    s1 = g["cases"]["stage1/bf16"]
    assert s1["T"] == 16 and "score1" not in s1 and s1["n_rows"] == synth.canonical_len(pkg.internvl2_8b(), 16) - 1
    gr = g["cases"]["greedy/bf16"]
    assert gr["tokens"].shape == (1, gr["n_new"]) and set(gr["tokens"][0].tolist()) <= set(gr["level_ids"])
    assert float(gr["margin_sigma"].min()) >= 0.99 * gr["margin_floor"]
    # the older fixture loads as plain data too
    g2 = torch.load(os.path.join(golden_dir, "e2e_8b_full.pt"), weights_only=True)
    assert "planted/201" in g2["cases"]
    # the second batch of the benched shape (make_golden_8b_r3b.py): same weights, inputs of seed 1
    g3 = torch.load(os.path.join(golden_dir, "e2e_8b_r3b.pt"), weights_only=True)
    assert g3["w_seed"] == g["w_seed"] and g3["overrides"] == g["overrides"]
    b3 = g3["cases"]["batch4/bf16"]
    assert (b3["B"], b3["T"], b3["seed"]) == (4, 8, 1) and b3["logit"].numel() == 40 and not torch.equal(b3["score1"], b4["score1"])


def _tiny_hf_lm(vocab: int, seed: int):
    from transformers import LlamaConfig, LlamaForCausalLM
    torch.manual_seed(seed)
    cfg = LlamaConfig(vocab_size=vocab, hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, max_position_embeddings=64, pad_token_id=0, bos_token_id=1, eos_token_id=2)
    m = LlamaForCausalLM(cfg).double().eval()
    with torch.no_grad():                      # sharper than the init's near-uniform logits, so that hypotheses differ in score
        m.lm_head.weight.mul_(12.0)
    return m


@pytest.mark.parametrize("num_beams,eos,length_penalty,early_stopping,max_new,procs", [
    (2, [], 1.0, False, 6, {}),
    (3, [2], 1.0, False, 8, {}),
    (4, [2, 5], 1.0, False, 7, {}),
    (3, [2], 2.0, False, 8, {}),
    (3, [2], 0.0, "never", 8, {}),
    (3, [2], 1.0, True, 8, {}),
    (4, [2], 0.6, True, 9, {}),
    (3, [2], 1.0, False, 8, {"repetition_penalty": 1.3}),
    (3, [2], 1.0, False, 8, {"no_repeat_ngram_size": 2}),
])
def test_beam_search_follows_transformers(num_beams, eos, length_penalty, early_stopping, max_new, procs):
    """beam.beam_search against transformers' own generate(num_beams > 1) on a small random causal LM, driven - as the reference drives its
    language model (modeling_internvl_chat.py:798-809) - through inputs_embeds, over several prompts, seeds and search settings: the same
    sequences, token for token.  The model is only a source of logits here (fp64, so that HF's cached forward and this test's re-computed
    one round to the same fp32 numbers)."""
    from aigv_assessor_amd import beam
    V, P, B = 11, 5, 3
    checked = 0
    for seed in range(4):
        lm = _tiny_hf_lm(V, seed)
        g = torch.Generator().manual_seed(100 + seed)
        prompt = torch.randint(3, V, (B, P), generator=g)
        emb = lm.get_input_embeddings()(prompt)
        kw = dict(max_new_tokens=max_new, num_beams=num_beams, do_sample=False, length_penalty=length_penalty, early_stopping=early_stopping,
                  eos_token_id=(eos if eos else None), pad_token_id=0, **procs)
        with torch.no_grad():
            want = lm.generate(inputs_embeds=emb, attention_mask=torch.ones(B, P, dtype=torch.long), **kw)
        # the same search over a re-computing step function: histories reordered instead of a KV cache
        hist = {"tok": torch.zeros((B, num_beams, 0), dtype=torch.long)}

        def logits_of(tok_hist):   # [B, nb, t] -> fp32 [B, nb, V]
            t = tok_hist.shape[2]
            e = emb[:, None].expand(B, num_beams, P, emb.shape[-1]).reshape(B * num_beams, P, -1)
            if t:
                e = torch.cat((e, lm.get_input_embeddings()(tok_hist.reshape(B * num_beams, t))), dim=1)
            with torch.no_grad():
                return lm(inputs_embeds=e).logits[:, -1, :].float().view(B, num_beams, V)

        def reorder(parent):
            hist["tok"] = torch.gather(hist["tok"], 1, parent[:, :, None].expand(-1, -1, hist["tok"].shape[2]))

        def step(tok):
            hist["tok"] = torch.cat((hist["tok"], tok[:, :, None]), dim=2)
            return logits_of(hist["tok"])

        processors = []
        if "repetition_penalty" in procs:
            processors.append(InternVLChatModel._repetition_penalty(procs["repetition_penalty"]))
        if "no_repeat_ngram_size" in procs:
            processors.append(InternVLChatModel._no_repeat_ngram(procs["no_repeat_ngram_size"]))
        first = logits_of(hist["tok"])[:, 0, :]
        got = beam.beam_search(first, step, reorder, num_beams, max_new, eos_ids=eos, pad_id=0, length_penalty=length_penalty,
                               early_stopping=early_stopping, processors=processors)
        assert got.shape == want.shape and torch.equal(got, want), (seed, got.tolist(), want.tolist())
        checked += 1
    assert checked == 4


def test_llama_family_config_and_streamed_repacking():
    """The reference's second LLM family (modeling_internvl_chat.py:228-233): a LlamaForCausalLM llm_config parses with transformers'
    LlamaConfig defaults where InternLM2's differ, the streamed re-packing equals the dict one, and other families still raise."""
    import aigv_assessor_amd as pkg
    from aigv_assessor_amd import synth, weights
    from aigv_assessor_amd.modeling import InternVLChatModel
    d = dict(architectures=["LlamaForCausalLM"], hidden_size=256, intermediate_size=384, num_attention_heads=4, num_key_value_heads=2, num_hidden_layers=2, vocab_size=64)
    cfg = pkg.InternVLChatConfig.from_dict(dict(llm_config=d, vision_config=dict(hidden_size=64, intermediate_size=128, num_attention_heads=1, num_hidden_layers=1)))
    l = cfg.llm_config
    assert l.architectures == ("LlamaForCausalLM",) and l.rope_theta == 10000.0 and l.rms_norm_eps == 1e-6 and l.head_dim == 64
    assert pkg.InternVLChatConfig.from_dict(dict(llm_config=dict(d, rope_parameters=dict(rope_theta=5e5, rope_type="default")))).llm_config.rope_theta == 5e5
    with pytest.raises(NotImplementedError):
        pkg.InternVLChatConfig.from_dict(dict(llm_config=dict(d, attention_bias=True)))
    packed = synth.make_state_dict(cfg, seed=2, rich=True)
    sd = weights.internlm2_to_llama(packed, l)
    assert len(sd) == len(packed) + 2 * l.num_hidden_layers and "language_model.model.layers.1.self_attn.k_proj.weight" in sd
    streamed = dict(weights.llama_stream_to_internlm2(iter(sd.items()), l))
    assert set(streamed) == set(packed) and all(torch.equal(streamed[k], packed[k]) for k in packed)
    with pytest.raises(KeyError):       # a layer whose v_proj never arrives
        dict(weights.llama_stream_to_internlm2(((k, v) for k, v in sd.items() if not k.endswith("layers.1.self_attn.v_proj.weight")), l))
    # rope scaling reaches the tables whatever the spelling, and what is not built raises instead of running unscaled
    base = dict(d, max_position_embeddings=4096)
    for given in (dict(rope_scaling=dict(type="linear", factor=2.0)), dict(rope_scaling=dict(rope_type="linear", factor=2.0)),
                  dict(rope_parameters=dict(rope_theta=10000.0, rope_type="linear", factor=2.0))):
        assert pkg.InternVLChatConfig.from_dict(dict(llm_config=dict(base, **given))).llm_config.rope_scaling == {"type": "linear", "factor": 2.0}
    assert pkg.InternVLChatConfig.from_dict(dict(llm_config=dict(base, rope_scaling=dict(rope_type="dynamic", factor=4)))).llm_config.rope_scaling["type"] == "dynamic"
    assert pkg.InternVLChatConfig.from_dict(dict(llm_config=dict(base, rope_scaling=dict(rope_type="default")))).llm_config.rope_scaling is None
    for bad in (dict(rope_scaling=dict(rope_type="llama3", factor=8.0)), dict(rope_parameters=dict(rope_type="yarn", factor=4.0)), dict(head_dim=32)):
        with pytest.raises(NotImplementedError):
            pkg.InternVLChatConfig.from_dict(dict(llm_config=dict(base, **bad)))
    lc = pkg.InternVLChatConfig.from_dict(dict(llm_config=dict(architectures=["LlamaForCausalLM"]))).llm_config      # LlamaConfig's own defaults
    assert (lc.max_position_embeddings, lc.vocab_size, lc.intermediate_size, lc.num_key_value_heads, lc.head_dim) == (2048, 32000, 11008, 32, 128)
    assert pkg.InternVLChatConfig.from_dict(dict(llm_config=dict(rope_scaling=dict(type="dynamic", factor=2.0)))).llm_config.rope_scaling == {"type": "dynamic", "factor": 2.0}
    m = InternVLChatModel(cfg)          # constructs (host side only) and owns InternLM2-layout parameters for both families
    assert m.llm_arch_name == "LlamaForCausalLM" and "language_model.model.layers.0.attention.wqkv.weight" in dict(m.named_parameters())
    # load_state_dict_stream == load_state_dict: Llama names and InternLM2 names both load on a Llama configuration, missing
    # tensors raise unless strict=False, an exception mid-stream still invalidates the native copy
    ref = InternVLChatModel(cfg)
    ref.load_state_dict(sd)
    for source in (sd, packed):
        m2 = InternVLChatModel(cfg)
        m2._dirty = False
        assert m2.load_state_dict_stream(iter(source.items())) == [] and m2._dirty
        assert all(torch.equal(a, b) for a, b in zip(m2.parameters(), ref.parameters()))
    m2 = InternVLChatModel(cfg)
    short = [(k, v) for k, v in packed.items() if not k.startswith("mlpscore.fc5")]
    with pytest.raises(RuntimeError, match="missing"):
        m2.load_state_dict_stream(iter(short))
    assert sorted(m2.load_state_dict_stream(iter(short), strict=False)) == ["mlpscore.fc5.bias", "mlpscore.fc5.weight"]
    m2._dirty = False
    with pytest.raises(RuntimeError, match="unexpected"):
        m2.load_state_dict_stream(iter(list(packed.items())[:3] + [("not.a.parameter", torch.zeros(1))]))
    assert m2._dirty
    cfg.llm_config.architectures = ("Qwen2ForCausalLM",)
    with pytest.raises(NotImplementedError):
        InternVLChatModel(cfg)


def test_llama_family_lora_adapters_merge_before_the_repacking():
    """A stage-2 LoRA checkpoint of the Llama family (peft targets q_proj / k_proj / v_proj / o_proj / gate_proj / up_proj / down_proj,
    modeling_internvl_chat.py:288-305): W + 2 B A on the HF names first, then the re-packing - equal to re-packing the merged matrices."""
    import aigv_assessor_amd as pkg
    from aigv_assessor_amd import synth, weights
    cfg = pkg.tiny(llm_hidden=128, llm_heads=2, llm_kv_heads=1, llm_layers=1, llm_inter=192, vocab=64)
    cfg.llm_config.architectures = ("LlamaForCausalLM",)
    l = cfg.llm_config
    sd = weights.internlm2_to_llama(synth.make_state_dict(cfg, seed=4, rich=True), l)
    g = torch.Generator().manual_seed(1)
    lora, want = {}, dict(sd)
    for name in ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj"):
        k = f"language_model.model.layers.0.{name}"
        w = sd[k + ".weight"]
        a, b = torch.randn(4, w.shape[1], generator=g).to(torch.bfloat16), torch.randn(w.shape[0], 4, generator=g).to(torch.bfloat16)
        lora[k.replace("language_model.", "language_model.base_model.model.") + ".lora_A.default.weight"] = a
        lora[k.replace("language_model.", "language_model.base_model.model.") + ".lora_B.default.weight"] = b
        want[k + ".weight"] = (w.float() + 2.0 * (b.float() @ a.float())).to(w.dtype)
    merged = weights.llama_to_internlm2(weights.merge_lora_state_dict(sd, lora), l)
    ref = weights.llama_to_internlm2(want, l)
    assert set(merged) == set(ref) and all(torch.equal(merged[k], ref[k]) for k in ref)
