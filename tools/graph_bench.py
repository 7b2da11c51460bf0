#!/usr/bin/env python3
"""GENConv aggregation kernels (csrc/graph.hip) alone on the step's block-diagonal 8-NN grid graph: bench.py's `genconv_roofline`
leg as its own program, for rocprofv3 (--kernel-trace --stats, or --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum passes).
usage: graph_bench.py [patches per bag = 4096] [bags = 16] [launches = 20]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from advmil_amd import ops  # noqa: E402

patches = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
bags = int(sys.argv[2]) if len(sys.argv) > 2 else 16
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
print(json.dumps(bench.genconv_roofline(torch, ops, torch.device("cuda:0"), patches, bags, iters)))
