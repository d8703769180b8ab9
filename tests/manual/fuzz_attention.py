"""Randomised run of the prefill attention kernel (aigv_op_attention) through the suite's own case (tests/test_gpu_ops.py::_attention_case: against fp64 truth and the eager bf16
evaluation, with a planted late maximum): random ragged length lists around the block edges (1, 31..33, 63..65, 127..129, 255..257, 1025, 2176, 2177) and uniformly up to 2600,
head configurations of InternViT (d = 64 / 128, non-causal) and InternLM2 (d = 128, causal, GQA groups 1 / 2 / 4 / 6), both score numerics, both kernel forms, the uniform-length hint.

    python tests/manual/fuzz_attention.py [n_cases = 60] [seed = 0]"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_ops as T  # noqa: E402
from aigv_assessor_amd import native  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lib = native.load()
rng = random.Random(seed0)
EDGES = [1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 513, 1025, 2176, 2177]
CONFIGS = [(64, False, 2, 2), (64, False, 3, 3), (128, False, 2, 2), (128, True, 4, 2), (128, True, 8, 2), (128, True, 2, 2), (128, True, 6, 1), (128, True, 4, 1)]
bad = 0
for c in range(n_cases):
    d, causal, h, hk = rng.choice(CONFIGS)
    lens = [rng.choice(EDGES) if rng.random() < 0.6 else rng.randint(1, 2600) for _ in range(rng.randint(1, 4))]
    if rng.random() < 0.25:
        lens = [lens[0]] * rng.randint(1, 3)                        # a uniform batch (the ViT shape of the hint)
    while sum(n * n for n in lens) * h > 4.0e7 and len(lens) > 1:   # the fp64 truth is O(n^2) on the host
        lens.pop()
    kernel, round_scores = rng.choice([0, 0, 8]), rng.random() < 0.5
    uniform = kernel == 0 and len(set(lens)) == 1 and rng.random() < 0.8
    T.sync(lib.aigv_tune_attention(kernel), lib)
    try:
        T._attention_case(lib, d, causal, h, hk, lens, uniform=uniform, round_scores=round_scores)
    except AssertionError as e:
        bad += 1
        print(f"CASE {c} FAILED: d {d} causal {causal} h {h} hk {hk} lens {lens} kernel {kernel} round_scores {round_scores} uniform {uniform}: {str(e)[:200]}", flush=True)
    finally:
        T.sync(lib.aigv_tune_attention(0), lib)
        T._KEEP.clear()
    if c % 10 == 9:
        print(f"case {c + 1}/{n_cases}: failed so far {bad}", flush=True)
assert bad == 0, bad
print(f"FUZZ_ATTN_OK {n_cases} cases")
