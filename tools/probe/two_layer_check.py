import os, sys, torch
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/advmil_amd") else os.getcwd())
from advmil_amd import ops
ops.set_gemm_mode("bf16x3")
dev="cuda:0"
def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1e3/iters
for M in (131072, 524288):
    K, N1, N2 = 1024, 384, 128
    x = torch.randn(M, K, device=dev); W1 = torch.randn(N1, K, device=dev)*0.05; W2 = torch.randn(N2, K, device=dev)*0.05
    b1 = torch.randn(N1, device=dev); b2 = torch.randn(N2, device=dev)
    xpl, p1, p2 = ops.split_planes(x), ops.split_planes(W1), ops.split_planes(W2)
    print("ok:", ops.gemm_two_layers_ok(M, N1, N2, K))
    y1, y2, cpl = ops.gemm_two_layers(x, xpl, W1, p1, b1, 1, W2, p2, b2, 0, True)
    c1 = ops.Planes(torch.empty(M, N1, dtype=torch.bfloat16, device=dev), torch.empty(M, N1, dtype=torch.bfloat16, device=dev))
    r1 = ops.gemm(x, W1, True, True, M, N1, K, bias=b1, act0=1, a_planes=xpl, b_planes=p1, c_planes=c1, splits=1)
    r2 = ops.gemm(x, W2, True, True, M, N2, K, bias=b2, act0=0, a_planes=xpl, b_planes=p2)
    print(M, "bit-identical:", torch.equal(y1, r1), torch.equal(y2, r2), torch.equal(cpl.hi, c1.hi), torch.equal(cpl.lo, c1.lo))
    t2 = bench(lambda: ops.gemm_two_layers(x, xpl, W1, p1, b1, 1, W2, p2, b2, 0, True))
    ta = bench(lambda: ops.gemm(x, W1, True, True, M, N1, K, bias=b1, act0=1, a_planes=xpl, b_planes=p1, c_planes=c1, splits=1))
    tb = bench(lambda: ops.gemm(x, W2, True, True, M, N2, K, bias=b2, act0=0, a_planes=xpl, b_planes=p2))
    print(f"  one launch {t2:.0f} us   separate {ta:.0f} + {tb:.0f} = {ta+tb:.0f} us")
M, K, N1, N2 = 131072, 1024, 384, 128
x = torch.randn(M, K, device=dev); W1 = torch.randn(N1, K, device=dev)*0.05; W2 = torch.randn(N2, K, device=dev)*0.05
b1 = torch.randn(N1, device=dev); b2 = torch.randn(N2, device=dev)
xpl, p1, p2 = ops.split_planes(x), ops.split_planes(W1), ops.split_planes(W2)
c1 = ops.Planes(torch.empty(M, N1, dtype=torch.bfloat16, device=dev), torch.empty(M, N1, dtype=torch.bfloat16, device=dev))
o1 = torch.empty(M, N1, device=dev)
print("t83 no planes, out fixed:", bench(lambda: ops.gemm(x, W1, True, True, M, N1, K, out=o1, bias=b1, act0=1, a_planes=xpl, b_planes=p1, splits=1)))
print("t83 + planes, out fixed :", bench(lambda: ops.gemm(x, W1, True, True, M, N1, K, out=o1, bias=b1, act0=1, a_planes=xpl, b_planes=p1, c_planes=c1, splits=1)))
print("t83 + planes, out alloc :", bench(lambda: ops.gemm(x, W1, True, True, M, N1, K, bias=b1, act0=1, a_planes=xpl, b_planes=p1, c_planes=c1, splits=1)))
print("two layers no planes    :", bench(lambda: ops.gemm_two_layers(x, xpl, W1, p1, b1, 1, W2, p2, b2, 0, False)))
print("two layers + planes     :", bench(lambda: ops.gemm_two_layers(x, xpl, W1, p1, b1, 1, W2, p2, b2, 0, True)))
