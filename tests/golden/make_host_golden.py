"""Golden vectors for the host-side input building, produced by the REFERENCE's own functions in this
container:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_host_golden.py        ->  tests/golden/host_inputs.pt

* ``preprocess_internlm`` (internvl/train/dataset.py:595-682) is imported from the reference (its module-level imports of
  cv2 / imageio / decord / torchvision, absent here and unused by this function, are satisfied by empty stand-in modules)
  and fed tests/stub_tokenizer.StubTokenizer - the real SentencePiece model ships with checkpoints only.
* ``get_index`` (stage2_eval.py:429-441) is a method of the eval script's dataset class; that script's import chain needs the
  HF Trainer / flash-attn / deepspeed stack, so the function's source is cut out of the reference file with ``ast`` at
  generation time and executed as it stands (numpy in scope) - the reference's code runs, only its outputs are stored.
* the ``internlm2-chat`` conversation template (internvl/conversation.py) renders the chat prompts.
The user-turn string of a video sample ("Frame1: <image>\\n ... Motion Feature: <image>" + question, stage2_eval.py:465-481) is an
INPUT here: ``video_get_item`` cannot run without decord and a video file.
"""
import ast
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ref_shims  # noqa: E402
from stub_tokenizer import StubTokenizer  # noqa: E402

PERSPECTIVES = [   # the four quality perspectives: question / answer strings in the form of shell/data/mydata_mos1_test.jsonl
    ("How would you rate the static quality of this video?", "The static quality of the video is good."),
    ("How would you rate the temporal smoothness of this video?", "The temporal smoothness of the video is poor."),
    ("How would you rate the dynamic degree of this video?", "The dynamic degree of the video is excellent."),
    ("How would you rate the text-video correspondence of this video?", " The text-video correspondence of the video is fair. "),
]


class _Any:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return self

    def __getattr__(self, n):
        return _Any()


def user_turn(question, n_frames):
    value = "<video>\n" + question
    special = "\n".join("Frame{}: <image>".format(i + 1) for i in range(n_frames)) + "\nMotion Feature: <image>"
    return value.replace("<video>\n", special)


def main():
    os.environ.setdefault("MASTER_PORT", "29578")
    import transformers  # noqa: F401
    for name, attrs in (("cv2", {}), ("imageio", {}), ("decord", dict(VideoReader=_Any, cpu=_Any)), ("torchvision", {}),
                        ("torchvision.transforms", dict(Compose=_Any, Lambda=_Any, Resize=_Any, ToTensor=_Any, Normalize=_Any)),
                        ("torchvision.transforms.functional", dict(InterpolationMode=types.SimpleNamespace(BICUBIC=3)))):
        ref_shims._stub(name, **attrs)
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    tiny_llm = dict(architectures=["InternLM2ForCausalLM"], hidden_size=4096, intermediate_size=512, num_attention_heads=32,
                    num_key_value_heads=8, num_hidden_layers=1, vocab_size=640)
    tiny_vis = dict(architectures=["InternVisionModel"], hidden_size=128, intermediate_size=256, num_attention_heads=2,
                    num_hidden_layers=1, image_size=448, patch_size=14)
    ref_shims.install(tiny_llm, tiny_vis)
    import internvl.train.dataset as D
    from internvl.conversation import get_conv_template

    out = {"samples": [], "get_index": [], "chat_prompts": {}}
    tok = StubTokenizer(92553)
    for (q, a), (T, ntok) in [(PERSPECTIVES[0], (8, 256)), (PERSPECTIVES[1], (8, 256)), (PERSPECTIVES[2], (8, 256)),
                              (PERSPECTIVES[3], (8, 256)), (PERSPECTIVES[0], (16, 256)), (PERSPECTIVES[1], (4, 64))]:
        conv = [{"from": "human", "value": user_turn(q, T)}, {"from": "gpt", "value": a}]
        counts = [ntok] * (T + 1)
        counts[-1] = 1
        ret = D.preprocess_internlm("internlm2-chat", [conv], tok, counts, group_by_length=True, ds_name="golden", num_image=T + 1)
        out["samples"].append(dict(question=q, answer=a, n_frames=T, num_image_token=ntok, input_ids=ret["input_ids"][0].clone(),
                                   labels=ret["labels"][0].clone(), attention_mask=ret["attention_mask"][0].clone()))
        n_ans = int((ret["labels"][0] != -100).sum())
        print(f"T={T} ntok={ntok}: N={ret['input_ids'].shape[1]}, {n_ans} label tokens: {tok.decode(ret['labels'][0][ret['labels'][0] != -100])!r}")

    # get_index: the function's own source, executed
    src_path = os.path.join(ref_shims.REF_ROOT, "internvl/train/internvl/eval/stage2_eval.py")
    with open(src_path) as f:
        tree = ast.parse(f.read())
    fn = next(n for c in ast.walk(tree) if isinstance(c, ast.ClassDef) and c.name == "LazySupervisedDataset"
              for n in c.body if isinstance(n, ast.FunctionDef) and n.name == "get_index")
    ns = {"np": np}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), src_path, "exec"), ns)
    for bound in (None, (0.0, 4.0), (1.5, 3.25)):
        for fps in (8.0, 23.976, 30.0):
            for max_frame in (7, 15, 48, 149, 1000):
                for seg in (8, 16, 4):
                    idx = ns["get_index"](None, bound, fps, max_frame, first_idx=0, num_segments=seg)
                    out["get_index"].append(dict(bound=bound, fps=fps, max_frame=max_frame, num_segments=seg, indices=[int(x) for x in idx]))
    print(len(out["get_index"]), "get_index cases; e.g.", out["get_index"][3])

    # chat prompts of the internlm2-chat template (modeling_internvl_chat.py:600-612)
    t = get_conv_template("internlm2-chat")
    t.append_message(t.roles[0], "<image>\nDescribe the quality.")
    t.append_message(t.roles[1], None)
    out["chat_prompts"]["single"] = t.get_prompt()
    t = get_conv_template("internlm2-chat")
    for q_, a_ in (("<image>\nDescribe the quality.", "It is fair."),):
        t.append_message(t.roles[0], q_)
        t.append_message(t.roles[1], a_)
    t.append_message(t.roles[0], "And the motion?")
    t.append_message(t.roles[1], None)
    out["chat_prompts"]["history"] = t.get_prompt()
    out["chat_prompts"]["sep"] = t.sep
    out["chat_prompts"]["system_message"] = t.system_message
    torch.save(out, os.path.join(HERE, "host_inputs.pt"))
    print("wrote", os.path.join(HERE, "host_inputs.pt"))


if __name__ == "__main__":
    main()
