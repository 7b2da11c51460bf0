import torch
dev="cuda:0"
def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1e3/iters
for mb in (64, 201, 403, 805):
    n = mb*1000*1000//4
    xs=[torch.empty(n, device=dev) for _ in range(4)]
    ys=[torch.randn(n, device=dev) for _ in range(4)]
    i=[0]
    def fill():
        i[0]=(i[0]+1)%4; xs[i[0]].fill_(1.0)
    def copy():
        i[0]=(i[0]+1)%4; xs[i[0]].copy_(ys[i[0]])
    def rd():
        i[0]=(i[0]+1)%4; return ys[i[0]].sum()
    t=bench(fill); print(f"{mb} MB fill: {t:.1f} us = {mb/t*1e-6*1e6/1e6:.2f} TB/s")
    t=bench(copy); print(f"{mb} MB copy: {t:.1f} us = {2*mb/t:.2f} TB/s (r+w)")
    t=bench(rd); print(f"{mb} MB sum : {t:.1f} us = {mb/t:.2f} TB/s")
