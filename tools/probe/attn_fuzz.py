#!/usr/bin/env python3
"""Randomised check of the fused attention core (ops.mha -> advmil_mha_fwd / advmil_mha_bwd) against float64: random numbers of ragged
bags (1 .. 700 tokens each, some up to 1 400, incl. 1-token bags and lengths around the 32 / 64 / 256-row tile boundaries), head_dim 16 / 32 / 48 / 64,
dropout 0 / 0.1 / 0.25 / 0.5 with the masks regenerated on the host from the kernels' counter hash, forward and all three gradient
blocks. usage: attn_fuzz.py [cases] [seed]"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from advmil_amd import ops  # noqa: E402
from tests.test_attention_gpu import NH, host_masks, ref_attention, relerr  # noqa: E402

dev = "cuda:0"
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
g = torch.Generator().manual_seed(rnd.randrange(1 << 30))
edge = (1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513)
worst = [0.0, 0.0]
for case in range(ncase):
    hd = rnd.choice((16, 32, 48, 48, 64))
    d = NH * hd
    nb = rnd.randint(1, 6)
    lens = [rnd.choice(edge) if rnd.random() < 0.4 else rnd.randint(1, 1400 if rnd.random() < 0.15 else 700) for _ in range(nb)]      # (up to six 256-key blocks: the single-pass backward's partial slabs)
    p = rnd.choice((0.0, 0.1, 0.25, 0.5))
    Lt = sum(lens)
    # (operand scale <= 1: the error of a split-bf16 score is ~2^-17 |q||k|, and the softmax turns an absolute score error into a
    # relative probability error -- at scale 1.5 / head_dim 16 the scores reach +-10 and the forward error 2e-5, still inside the 1e-4
    # contract; the ESAT layer's scores are O(1))
    qkv = torch.randn(Lt, 3 * d, generator=g) * rnd.choice((0.3, 0.7, 1.0))
    go = torch.randn(Lt, d, generator=g)
    seg = ops.Segments(lens, dev) if (nb > 1 or rnd.random() < 0.5) else None
    seed = rnd.randrange(1, 1 << 20)
    rng = ops.DeviceRng(dev, seed=seed)
    rng.record = True
    a = qkv.clone().to(dev).requires_grad_(True)
    o = ops.mha(a, NH, p, rng, seg=seg)
    (o * go.to(dev)).sum().backward()
    masks = None
    if p > 0:
        (_, sid, _, _), = [e for e in rng.log if e[0] == "mha_attn"]
        masks = host_masks(seed, sid, lens, p)
        if p != 0.25:                                 # the kernel quantises p to 1/256 (0.25 is exact): the scale is 256 / (256 - floor(256 p))
            from advmil_amd import synth
            sc = synth.attn_dropout_scale(p) * (1.0 - p)
            masks = [m * sc for m in masks]
    r = qkv.clone().double().requires_grad_(True)
    orf = ref_attention(r, lens, masks, HD=hd)
    (orf * go.double()).sum().backward()
    e_f = relerr(o, orf)
    # backward: per bag, against that bag's own gradient scale (max over its q | k | v blocks). A 1- or 2-token bag has gradients
    # of O(|dO||v|) ~ 5 in dv and EXACT zeros in dq / dk (softmax over one key is constant): the split-bf16 products leave ~2^-17 of
    # that scale there (1-2e-5 absolute), which a per-block maximum taken over all bags (0.2 for the long bags) would misread
    e_b, r0 = 0.0, 0
    for L in lens:
        ga, gr = a.grad[r0:r0 + L].detach().cpu().double(), r.grad[r0:r0 + L]
        e_b = max(e_b, float((ga - gr).abs().max() / (gr.abs().max() + 1e-30)))
        r0 += L
    worst = [max(worst[0], e_f), max(worst[1], e_b)]
    # (split-bf16 products carry 2^-17 = 7.6e-6 each; the unit tests' 1e-5 holds at p <= 0.25, at p = 0.5 the x2 rescale of half as
    # many kept terms reaches 1.1e-5 of the output's maximum. Round 4, seed 2 case 4 -- head_dim 48, six ~500-token bags, operand scale
    # 1.0 (scores up to +-5), p = 0.25 -- measured 2.0e-5 forward on kernels whose arithmetic had not changed since the round-3 run
    # (worst 1.7e-5 over seed 1): the bound is 3e-5 here, the contract is 1e-4)
    ok = e_f < 3e-5 and e_b < 5e-5 and bool(torch.isfinite(o).all()) and bool(torch.isfinite(a.grad).all())
    note = ""
    if (not ok and e_f < 1e-4 and e_b < 5e-5) or os.environ.get("ATTN_FUZZ_MODEL") == str(case):
        # Is a forward deviation beyond 3e-5 the ARITHMETIC's (split-bf16: hi.hi + hi.lo + lo.hi per product, the lo.lo term dropped) or
        # a kernel's? The same forward in float64 with exactly those terms dropped -- scores from the split q / k, output from the split
        # P / V. A case whose kernel output follows that model to 1.5e-5 while staying inside the 1e-4 contract is the arithmetic at work
        # (large scaled scores: seeds 702 / 705, head_dim 16, |s| up to 5.8 / 7.3) and is accepted, with the three numbers printed.
        def split(x):
            hi = x.float().bfloat16().double()
            lo = (x - hi).float().bfloat16().double()
            return hi, lo
        outs, r0, smax = [], 0, 0.0
        D_ = NH * hd
        for b, L in enumerate(lens):
            blk = qkv[r0:r0 + L].double()
            q_, k_, v_ = (t.reshape(L, NH, hd).transpose(0, 1) for t in blk.split(D_, dim=1))
            smax = max(smax, float((q_ @ k_.transpose(-1, -2)).abs().max() / hd ** 0.5))
            (qh, ql), (kh, kl), (vh, vl) = split(q_), split(k_), split(v_)
            s_ = (qh @ kh.transpose(-1, -2) + qh @ kl.transpose(-1, -2) + ql @ kh.transpose(-1, -2)) / hd ** 0.5
            pr = torch.softmax(s_, dim=-1)
            if masks is not None:
                pr = pr * masks[b]
            ph, pl = split(pr)
            outs.append((ph @ vh + ph @ vl + pl @ vh).transpose(0, 1).reshape(L, D_))
            r0 += L
        om = torch.cat(outs, dim=0)
        e_model, e_km = relerr(om, orf), relerr(o, om)
        note = f" [largest |scaled score| {smax:.1f}; split-bf16 model vs float64 {e_model:.2e}; kernel vs the model {e_km:.2e}]"
        if not ok and e_km < 1.5e-5 and bool(torch.isfinite(o).all()) and bool(torch.isfinite(a.grad).all()):
            ok = True
            note += " accepted: the arithmetic's deviation, inside the 1e-4 contract"
    print(f"case {case}: head_dim {hd} bags {lens} p {p} seg {'yes' if seg is not None else 'no'}: fwd {e_f:.1e} bwd {e_b:.1e} {'ok' if ok else 'FAIL'}{note}", flush=True)
    if not ok:
        sys.exit(1)
print("all ok; worst fwd / bwd relative error", worst)
