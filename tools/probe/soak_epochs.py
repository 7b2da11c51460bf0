#!/usr/bin/env python3
"""Soak: several epochs of the product loop + per-epoch evaluation over a few hundred distinct ragged bags; per epoch: bags/s, device
memory allocated / reserved, cache statistics. Looks for leaks, allocator growth and slow-downs. usage: soak_epochs.py [epochs] [patients]"""
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from advmil_amd import ingest  # noqa: E402
from advmil_amd.config import default_cfg  # noqa: E402
from advmil_amd.model import MyHandler  # noqa: E402

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 5
npat = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda", 0)
rnd = random.Random(5)
g = torch.Generator().manual_seed(5)
base = [torch.randn(1, 12288, 1024, generator=g) for _ in range(8)]            # 8 pageable source bags, sliced to ragged lengths
lens = [16 * rnd.randint(128, 768) for _ in range(npat)]


class DS:
    def __init__(self, idx):
        self.idx = idx


class DL:
    def __init__(self, idx, shuffle):
        self.dataset, self.shuffle = DS(idx), shuffle

    def __iter__(self):
        order = list(self.dataset.idx)
        if self.shuffle:
            rnd.shuffle(order)
        for i in order:
            yield (torch.tensor([[i]], dtype=torch.int), [base[i % 8][:, :lens[i]], torch.zeros(1, 1)], torch.tensor([[0.3 + 0.001 * i, float(i % 2)]]))


train, val = DL(list(range(npat)), True), DL(list(range(npat // 4)), False)
h = MyHandler(default_cfg(bcb_mode="abmil", bp_every_batch=16, cuda_id=0, gemm_mode="bf16x3"), device=dev)
h.patient_id.update({"train": [str(i) for i in range(npat)], "label_visible": [str(i) for i in range(npat)]})
for ep in range(epochs):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cl = h._train_each_epoch(train, "train")
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    ev = MyHandler.test_model(h.netG, h.netD, "abmil", val, times_test_sample=1)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    logs = h.pop_logs()
    assert bool(torch.isfinite(cl["y_hat"]).all()) and bool(torch.isfinite(ev["f_fake"]).all())
    st = ingest.device_bag_cache(dev).stats()
    print(f"epoch {ep}: train {cl['y'].shape[0] / (t1 - t0):7.0f} bags/s, eval {ev['y'].shape[0] / (t2 - t1):7.0f} bags/s, "
          f"allocated {torch.cuda.memory_allocated() / 1e9:6.2f} GB, reserved {torch.cuda.memory_reserved() / 1e9:6.2f} GB, cache {st}, "
          f"loss_D {logs[-2]['train_batch/netD/Loss_D']:.4f}", flush=True)
