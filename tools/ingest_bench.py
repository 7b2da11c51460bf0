#!/usr/bin/env python3
"""PCIe-inclusive rate: `_train_each_epoch` fed CPU bags (pageable, as a DataLoader yields them) through the staging slab
(advmil_amd/ingest.py), eager launches. Compare with bench.py's HBM-resident number."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from advmil_amd.config import default_cfg
from advmil_amd.model import MyHandler
N, BAGS, STEPS = 8192, 16, int(sys.argv[1]) if len(sys.argv) > 1 else 6
h = MyHandler(default_cfg(bp_every_batch=BAGS), device="cuda:0")
g = torch.Generator().manual_seed(0)
pool = [torch.randn(1, N, 1024, generator=g) for _ in range(32)]
if len(sys.argv) > 2 and sys.argv[2] == "pinned":
    pool = [t.pin_memory() for t in pool]
loader = [(torch.tensor([[i]], dtype=torch.int), [pool[i % 32], torch.zeros(1, 1)], torch.tensor([[0.5, float(i % 2)]])) for i in range(BAGS * (STEPS + 2))]
h._train_each_epoch(loader[:2 * BAGS], "train")          # warm-up: allocates the pinned + device slabs
torch.cuda.synchronize()
t0 = time.perf_counter()
h._train_each_epoch(loader[2 * BAGS:], "train")
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"ingest-inclusive: {BAGS * STEPS / dt:.1f} bags/s ({1e3 * dt / STEPS:.2f} ms/step, {BAGS * N * 4096 / 1e6:.0f} MB H2D per step, eager, {'pinned' if pool[0].is_pinned() else 'pageable'} loader tensors)")
