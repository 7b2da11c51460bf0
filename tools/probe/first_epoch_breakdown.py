import os, sys, time, random, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from advmil_amd import ingest, ops
from advmil_amd.config import default_cfg
from advmil_amd.model import MyHandler
dev = torch.device("cuda", 0)
rnd = random.Random(5); g = torch.Generator().manual_seed(5)
base = [torch.randn(1, 12288, 1024, generator=g) for _ in range(8)]
npat = 128
lens = [16 * rnd.randint(128, 768) for _ in range(npat)]
items = [(torch.tensor([[i]], dtype=torch.int), [base[i % 8][:, :lens[i]], torch.zeros(1, 1)], torch.tensor([[0.3, float(i % 2)]])) for i in range(npat)]
h = MyHandler(default_cfg(bcb_mode="abmil", bp_every_batch=16, cuda_id=0, gemm_mode="bf16x3"), device=dev)
acc = {}
def wrap(obj, name, label=None):
    f = getattr(obj, name); label = label or name
    def gfn(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); acc[label] = acc.get(label, 0.0) + time.perf_counter() - t; return r
    setattr(obj, name, gfn)
for nm in ("begin", "add", "add_device", "ready", "release", "pad_rows", "_ensure", "_ensure_planes"): wrap(ingest.SlabStager, nm, "stager." + nm)
wrap(ingest.BagCache, "put", "cache.put"); wrap(ingest, "host_copy_rows")
for nm in ("_plan", "_update_disc", "_update_gen"): wrap(MyHandler, nm)
for ep in range(2):
    acc.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    h._train_each_epoch(items, "train")
    th = time.perf_counter() - t0
    torch.cuda.synchronize(); ta = time.perf_counter() - t0
    print(f"epoch {ep}: host {1e3*th:.0f} ms, wall {1e3*ta:.0f} ms for {npat} bags; pieces (ms):", {k: round(1e3*v) for k, v in sorted(acc.items(), key=lambda kv: -kv[1])})
