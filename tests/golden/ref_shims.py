"""Test-harness only: import the *reference* AIGV-Assessor modules in this container.

The reference (/root/reference, read-only, never shipped) is pure Python but needs a few
third-party modules that are absent here and opens a hard-coded absolute path.  This file
installs the four shims described in SURVEY.md §8c so that ``make_golden.py`` can run the
reference's own CPU path and record golden vectors.  Nothing here is imported by the product
path, by ``-m gpu`` tests, by ``smoke()`` or by ``bench.py``.
"""
import importlib.machinery
import io
import json
import os
import sys
import types

REF_ROOT = os.environ.get("AIGV_REFERENCE_ROOT", "/root/reference")


def _stub(name, **attrs):
    mod = types.ModuleType(name)
    mod.__spec__ = importlib.machinery.ModuleSpec(name, loader=None)
    mod.__path__ = []
    for k, v in attrs.items():
        setattr(mod, k, v)
    sys.modules[name] = mod
    return mod


def install(llm_config: dict, vision_config: dict):
    """Install shims and return the reference's eval2 / eval1 modeling modules."""
    sys.dont_write_bytecode = True
    import torch
    import transformers  # noqa: F401  (must be imported before the stubs exist)

    class _Identity(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def forward(self, x):
            return x

    # shim 1: absent third-party modules
    _stub("timm")
    _stub("timm.models")
    _stub("timm.models.layers", DropPath=_Identity)

    def _no_slowfast(*a, **k):
        raise RuntimeError("slowfast_r50 needs a network download; stubbed")

    _stub("pytorchvideo")
    _stub("pytorchvideo.models")
    _stub("pytorchvideo.models.hub", slowfast_r50=_no_slowfast)
    _stub("peft", LoraConfig=object, get_peft_model=lambda m, c: m)

    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)

    # shim 2: the hard-coded /DATA/... config open in InternVLChatConfig.__init__
    import internvl.model.internvl_chat_eval2.configuration_internvl_chat as cfg2
    import internvl.model.internvl_chat_eval1.configuration_internvl_chat as cfg1

    payload = json.dumps({"llm_config": llm_config, "vision_config": vision_config})

    def fake_open(path, *a, **k):
        if str(path).startswith("/DATA/"):
            return io.StringIO(payload)
        return open(path, *a, **k)

    cfg2.open = fake_open
    cfg1.open = fake_open

    import internvl.model.internvl_chat_eval2.modeling_internvl_chat as m2
    import internvl.model.internvl_chat_eval1.modeling_internvl_chat as m1

    # shim 3: SlowFast stand-in -> motion feature becomes a captured INPUT
    class SlowFastStandIn(torch.nn.Module):
        feature = None  # set by the caller: [B, 2304]

        def forward(self, x):
            f = type(self).feature
            return f.reshape(f.shape[0], 2304, 1, 1, 1)

    m2.slowfast = SlowFastStandIn
    if hasattr(m1, "slowfast"):
        m1.slowfast = SlowFastStandIn

    # shim 4: forward() calls torch.distributed.get_rank()
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=0, world_size=1)
    return m2, m1, cfg2, SlowFastStandIn
