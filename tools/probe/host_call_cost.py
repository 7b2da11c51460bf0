"""Host cost of one launch through the Python wrappers (eager product loop: ~165 launches per step, host issue 3.6 ms).
usage: python tools/probe/host_call_cost.py"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from advmil_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
A = torch.randn(32, 128, device=dev)
B = torch.randn(128, 128, device=dev)
bias = torch.randn(128, device=dev)
out = torch.empty(32, 128, device=dev)
N = 3000


def t(fn, n=N):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return 1e6 * dt / n


print("ops.gemm (alloc out)          %.1f us" % t(lambda: ops.gemm(A, B, True, True, 32, 128, 128, bias=bias, act0=1)))
print("ops.gemm (out given)          %.1f us" % t(lambda: ops.gemm(A, B, True, True, 32, 128, 128, out=out, bias=bias, act0=1)))
e = ops.Epilogue()
e.alpha = 1.0
e.act_split = 1 << 30
L = _lib.lib()
st = ops._stream()
pa, pb, po = ops._p(A), ops._p(B), ops._p(out)
print("raw ctypes advmil_gemm_f32_tiled   %.1f us" % t(lambda: L.advmil_gemm_f32_tiled(1, 1, 32, 128, 128, pa, 128, pb, 128, po, 128, ctypes.byref(e), 1, 11, None, 0, st)))
print("ops._stream()                 %.2f us" % t(lambda: ops._stream()))
print("ops.Epilogue()                %.2f us" % t(lambda: ops.Epilogue()))
print("torch.empty(32,128)           %.2f us" % t(lambda: torch.empty(32, 128, dtype=torch.float32, device=dev)))
print("A.data_ptr()                  %.2f us" % t(lambda: A.data_ptr()))
print("ops._p(A)                     %.2f us" % t(lambda: ops._p(A)))
print("ops._chk(A)                   %.2f us" % t(lambda: ops._chk(A, 'A')))
print("torch.mm                      %.1f us" % t(lambda: torch.mm(A, B)))
x = torch.randn(64, 128, device=dev, requires_grad=True)
W = torch.randn(128, 128, device=dev, requires_grad=True)
bb = torch.randn(128, device=dev, requires_grad=True)


def fb():
    y = ops.linear_act(x, W, bb, act="relu")
    y.backward(y)


try:
    print("ops.linear fwd+bwd (autograd) %.1f us" % t(fb, 500))
except Exception as exc:
    print("linear probe skipped:", exc)

# where do the 386 us go? forward alone, backward() alone, and the Python body of LinearActFn.backward (runs on the engine's thread)
acc = {"py": 0.0, "n": 0}
orig = ops.LinearActFn.backward


def timed_bwd(ctx, *g):
    t0 = time.perf_counter()
    r = orig(ctx, *g)
    acc["py"] += time.perf_counter() - t0
    acc["n"] += 1
    return r


ops.LinearActFn.backward = staticmethod(timed_bwd)
tf = tb = 0.0
for i in range(600):
    if i == 100:
        tf = tb = 0.0
        acc["py"] = 0.0
        acc["n"] = 0
    t0 = time.perf_counter()
    y = ops.linear_act(x, W, bb, act="relu")
    t1 = time.perf_counter()
    y.backward(y)
    t2 = time.perf_counter()
    tf += t1 - t0
    tb += t2 - t1
torch.cuda.synchronize()
print("linear_act forward %.1f us, backward() %.1f us of which LinearActFn.backward body %.1f us (%d calls)" %
      (1e6 * tf / 500, 1e6 * tb / 500, 1e6 * acc["py"] / max(acc["n"], 1), acc["n"]))
import cProfile
import pstats
gy = torch.randn(64, 128, device=dev)
pr = cProfile.Profile()
y = ops.linear_act(x, W, bb, act="relu")
ctxs = []
# run the backward body on THIS thread under cProfile through a tiny autograd graph evaluated with torch.autograd.backward is not
# possible (engine thread); emulate by calling forward under profile only
pr.enable()
for _ in range(300):
    y = ops.linear_act(x, W, bb, act="relu")
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
