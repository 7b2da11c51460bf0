#!/usr/bin/env python3
"""Randomised world-size invariance: two ranks sharing the GPU (gloo) against one process over the same global step batches, shipped
dropout ON, random ragged bag lengths (multiples of 16; half of the cases with slabs >= 4096 rows per rank so that each rank pads its
own slab), through tests/test_parallel_gpu.py's comparison. usage: dp_fuzz.py [cases] [seed]"""
import os
import pathlib
import random
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
from tests.test_parallel_gpu import test_multi_rank_step_equals_single_rank_with_dropout_on as check  # noqa: E402

ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
for case in range(ncase):
    kind = rnd.choice(("abmil", "patch", "cluster"))
    big = rnd.random() < 0.5
    lens = [16 * rnd.randint(1, 40) for _ in range(8)]
    if big:
        lens = [n + 16 * rnd.randint(120, 200) for n in lens]
    os.environ["DP_LENS"] = ",".join(str(n) for n in lens)
    with tempfile.TemporaryDirectory() as d:
        check(kind + "-env", 2, pathlib.Path(d))
    print(f"case {case}: {kind} lens {lens}: ok", flush=True)
print("all ok")
