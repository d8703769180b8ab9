"""Randomised check of the prefill attention op (aigv_op_attention) on an MI355X - by hand, not part of the suites:

    python tests/manual/fuzz_attention.py [cases] [seed]

Per case: random packed sequences (1 .. 1400 rows; tile-boundary lengths, length-1 sequences, 64 j + 1 keys), d in {64, 128}, causal or not,
heads / kv heads with group sizes 1, 2, 4, both score numerics, the uniform-length hint where it applies, the lead-key form where it applies -
against fp64 truth with the suite's bars (tests/test_gpu_ops.py::_attention_case: at least as accurate as the reference's eager bf16 arithmetic)."""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import test_gpu_ops as T  # noqa: E402
from aigv_assessor_amd import native  # noqa: E402

lib = native.load()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for ci in range(cases):
    d = rng.choice([64, 128])
    causal = rng.random() < 0.5
    hk = rng.choice([1, 2, 3])
    h = hk * rng.choice([1, 2, 4])
    special = [1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 513, 1024, 1025, 1153]
    n_seq = rng.randint(1, 4)
    if rng.random() < 0.3:
        lens = [rng.choice([257, 1025, 641, 128])] * n_seq
    else:
        lens = [rng.choice(special) if rng.random() < 0.5 else rng.randint(1, 1400) for _ in range(n_seq)]
    rs = rng.random() < 0.5
    uniform = len(set(lens)) == 1 and rng.random() < 0.7
    tag = f"case {ci}: d {d} causal {causal} heads {h}/{hk} lens {lens} round_scores {rs} uniform {uniform}"
    try:
        T._attention_case(lib, d, causal, h, hk, lens, uniform=uniform, round_scores=rs)
        print("ok  ", tag, flush=True)
    except AssertionError as e:
        bad += 1
        print("FAIL", tag, "->", str(e)[:300], flush=True)
print(f"{cases - bad} / {cases} cases clean")
sys.exit(1 if bad else 0)
