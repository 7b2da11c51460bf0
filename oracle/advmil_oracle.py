"""ORACLE — TEST INFRASTRUCTURE ONLY. Not product code.

A plain-PyTorch CPU fp32 restatement of the AdvMIL generator+discriminator training path
(liupei101/AdvMIL @ v1). Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg may import this module; the product (`advmil_amd/`) never does and
fails loudly when its HIP library is missing.

Every function cites the reference file:line it restates. All randomness is an explicit
input: `masks` are the multiplicative dropout tensors (keep/(1-p), or None = identity) and
`noise` the generator noise tensors, so a HIP run with in-kernel counter RNG can be replayed
here bit-for-meaning (`advmil_amd.synth.dropout_keep` restates the kernel's decision).

Pinning: `tests/golden/gen_golden.py` imports the real reference (with import shims) in the
build container, checks this restatement against it (<= 1e-6) and writes the golden vectors
under `tests/golden/`; `tests/test_oracle_golden.py` re-checks the oracle against those
vectors everywhere. PatchGCN/GENConv is the exception: its arithmetic lives in an absent,
un-pinned torch_geometric => "parity unpinned" for `patch_gcn` below.

Parameters are passed as flat dicts with the reference's own state_dict key names.
"""
import math

import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------
# helpers
# ---------------------------------------------------------------------------------------
def _lin(P, name, x):
    return x @ P[name + ".weight"].t() + P[name + ".bias"]


def _drop(x, masks, key):
    if masks is None:
        return x
    m = masks.get(key)
    return x if m is None else x * m


def _sub(P, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in P.items() if k.startswith(prefix)}


# ---------------------------------------------------------------------------------------
# a1 Attn_Net_Gated — model/backbone_utils.py:11-29
# ---------------------------------------------------------------------------------------
def attn_net_gated(P, h, masks=None, mkey=""):
    """A[N,1] = attention_c(tanh(attention_a h) * sigmoid(attention_b h)); Dropout(.25) on both
    branches in train mode (backbone_utils.py:17-19)."""
    a = _drop(torch.tanh(_lin(P, "attention_a.0", h)), masks, mkey + "att_a")
    b = _drop(torch.sigmoid(_lin(P, "attention_b.0", h)), masks, mkey + "att_b")
    return _lin(P, "attention_c", a * b)


# ---------------------------------------------------------------------------------------
# a2 ABMIL — model/backbone.py:54-86
# ---------------------------------------------------------------------------------------
def abmil(P, x, masks=None):
    """x[1,N,C] -> (H[1,dim_out], A[1,N] softmax attention). backbone.py:79-86."""
    x = x.squeeze(0)
    h = _drop(torch.relu(_lin(P, "attention_net.0", x)), masks, "fc")
    s = attn_net_gated(_sub(P, "attention_net.3."), h, masks)          # [N,1]
    A = torch.softmax(s.t(), dim=1)                                    # [1,N]
    pooled = A @ h                                                     # [1,D]
    H = _drop(torch.relu(_lin(P, "rho.0", pooled)), masks, "rho")
    return H, A


# ---------------------------------------------------------------------------------------
# a3 DeepAttMISL — model/backbone.py:89-123
# ---------------------------------------------------------------------------------------
def deep_att_misl(P, x, cluster_id, masks=None, num_clusters=8):
    """phis is a 1x1 conv == FC+ReLU per patch; per-cluster mean; empty cluster -> zeros
    (backbone.py:112-116); then FC+ReLU+Dropout -> gated attention over the 8 rows -> mm."""
    x = x.squeeze(0)
    cid = cluster_id.reshape(-1)
    W = P["phis.0.weight"].reshape(P["phis.0.weight"].shape[0], -1)
    b = P["phis.0.bias"]
    rows = []
    for c in range(num_clusters):
        xc = x[cid == c]
        if xc.shape[0] == 0:
            rows.append(torch.zeros(W.shape[0], dtype=x.dtype))
        else:
            rows.append(torch.relu(xc @ W.t() + b).mean(dim=0))
    hc = torch.stack(rows, dim=0)                                      # [8,D]
    h = _drop(torch.relu(_lin(P, "attention_net.0", hc)), masks, "fc")
    s = attn_net_gated(_sub(P, "attention_net.3."), h, masks)
    A = torch.softmax(s.t(), dim=1)
    return A @ h, A


# ---------------------------------------------------------------------------------------
# a4 GAPool — model/backbone_utils.py:31-56 (pools its INPUT)
# ---------------------------------------------------------------------------------------
def gapool(P, x, masks=None, mkey="pool_"):
    """x[B,L,d] -> (out[B,d], attn[B,1,L])."""
    emb = _drop(torch.tanh(_lin(P, "fc1.0", x)), masks, mkey + "a")
    scr = _drop(torch.sigmoid(_lin(P, "score.0", x)), masks, mkey + "b")
    rep = _lin(P, "fc2", emb * scr).transpose(2, 1)                    # [B,1,L]
    attn = torch.softmax(rep, dim=2)
    return torch.matmul(attn, x).squeeze(1), attn


# ---------------------------------------------------------------------------------------
# a5 AVGPoolPatchEmbedding — model/backbone_utils.py:129-168 (+ sequence2square 62-77)
# ---------------------------------------------------------------------------------------
def avgpool_patch_embedding(P, x):
    """ksize=1, stride=1, scale=4: the 1x1 conv on 4x4 tiles is a row-wise FC; then
    LayerNorm(d, eps 1e-5) -> ReLU -> mean over each consecutive 16 rows. N % 16 == 0
    (backbone_utils.py:65)."""
    B, N, C = x.shape
    assert N % 16 == 0
    W = P["conv.weight"].reshape(P["conv.weight"].shape[0], -1)
    y = x @ W.t() + P["conv.bias"]
    y = F.layer_norm(y, (W.shape[0],), P["norm.weight"], P["norm.bias"], 1e-5)
    y = torch.relu(y)
    return y.reshape(B, N // 16, 16, W.shape[0]).mean(dim=2)


# ---------------------------------------------------------------------------------------
# a6 TransformerEncoderLayer (post-norm, relu, batch_first) — backbone_utils.py:113-127;
# arithmetic is torch's nn.TransformerEncoderLayer / nn.MultiheadAttention (third party).
# ---------------------------------------------------------------------------------------
def transformer_encoder_layer(P, x, nhead=8, masks=None):
    """x[1,L,d]. masks: 'attn' [1,H,L,L] on the softmax probabilities, 'drop1'/'drop2' on the
    two residual branches, 'ffn' on relu(linear1)."""
    B, L, d = x.shape
    hd = d // nhead
    qkv = x @ P["self_attn.in_proj_weight"].t() + P["self_attn.in_proj_bias"]
    q, k, v = qkv.chunk(3, dim=-1)
    q = q.reshape(B, L, nhead, hd).transpose(1, 2)
    k = k.reshape(B, L, nhead, hd).transpose(1, 2)
    v = v.reshape(B, L, nhead, hd).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
    p = _drop(torch.softmax(s, dim=-1), masks, "attn")
    o = (p @ v).transpose(1, 2).reshape(B, L, d)
    o = _lin(P, "self_attn.out_proj", o)
    x = F.layer_norm(x + _drop(o, masks, "drop1"), (d,), P["norm1.weight"], P["norm1.bias"], 1e-5)
    f = _drop(torch.relu(_lin(P, "linear1", x)), masks, "ffn")
    f = _lin(P, "linear2", f)
    x = F.layer_norm(x + _drop(f, masks, "drop2"), (d,), P["norm2.weight"], P["norm2.bias"], 1e-5)
    return x


# ---------------------------------------------------------------------------------------
# a7 DualTrans_HS (ESAT) — model/backbone.py:171-196 (coord=None: PE skipped,
# model_handler.py:390)
# ---------------------------------------------------------------------------------------
def dualtrans_hs(P, x, masks=None, nhead=8):
    emb = avgpool_patch_embedding(_sub(P, "patch_embedding_layer."), x)
    feat = transformer_encoder_layer(_sub(P, "patch_encoder_layer.layers.0."), emb, nhead, masks)
    H, attn = gapool(_sub(P, "pool."), feat, masks)
    return H, attn.squeeze(1)


# ---------------------------------------------------------------------------------------
# a15 PatchGCN — model/backbone.py:126-168.  PARITY UNPINNED: GENConv's arithmetic is
# torch_geometric's (absent, unpinned). Restated from the published PyG>=1.6 GENConv:
# message relu(x_j)+1e-7, per-channel scatter-softmax(msg*t) over in-edges at the target,
# sum, + x_dst, then MLP [d, 2d, d] with LayerNorm+ReLU between.
# ---------------------------------------------------------------------------------------
def genconv(P, x, edge_index, eps=1e-7):
    src, dst = edge_index[0], edge_index[1]
    n, d = x.shape
    msg = torch.relu(x[src]) + eps
    z = msg * P["t"]
    zmax = torch.full((n, d), -float("inf"), dtype=x.dtype).scatter_reduce(
        0, dst[:, None].expand(-1, d), z, reduce="amax", include_self=True)
    e = torch.exp(z - zmax[dst])
    den = torch.zeros(n, d, dtype=x.dtype).index_add_(0, dst, e)
    w = e / den[dst]
    out = torch.zeros(n, d, dtype=x.dtype).index_add_(0, dst, w * msg) + x
    h = _lin(P, "mlp.0", out)
    h = torch.relu(F.layer_norm(h, (h.shape[-1],), P["mlp.1.weight"], P["mlp.1.bias"], 1e-5))
    return _lin(P, "mlp.4", h)


def patch_gcn(P, x, edge_index, masks=None):
    """num_layers=1: only layers[0].conv runs (backbone.py:157); cat -> path_phi -> gated pool."""
    h0 = _drop(torch.relu(_lin(P, "fc.0", x)), masks, "fc")
    h1 = genconv(_sub(P, "layers.0.conv."), h0, edge_index)
    h = torch.cat([h0, h1], dim=1)
    h = _drop(torch.relu(_lin(P, "path_phi.0", h)), masks, "phi")
    s = attn_net_gated(_sub(P, "path_attention_head."), h, masks)
    A = torch.softmax(s.t(), dim=1)
    return A @ h, A


# ---------------------------------------------------------------------------------------
# a8 EmbedXLayer — model/model_utils.py:188-210; make_efficient_mlp_layer 157-166
# ---------------------------------------------------------------------------------------
def _eff_mlp(P, x, masks, key):
    h = _drop(torch.relu(_lin(P, "0", x)), masks, key)
    return _lin(P, "3", h)


def embed_x_layer(P, x, masks=None):
    """x[1,N,C] -> (fc_bag[1,d], fc_ins[1,L,d], pool attention[1,L])."""
    emb_ins = avgpool_patch_embedding(_sub(P, "embedding."), x)
    fc_ins = _eff_mlp(_sub(P, "fc1."), emb_ins, masks, "fc1")
    emb_bag, attn = gapool(_sub(P, "pool."), fc_ins, masks)
    fc_bag = _eff_mlp(_sub(P, "fc2."), emb_bag, masks, "fc2")
    return fc_bag, fc_ins, attn.squeeze(1)


# ---------------------------------------------------------------------------------------
# a9 make_embedding_y_layer — model/model_utils.py:168-186 (norm False, dropout 0.0 in cfg)
# ---------------------------------------------------------------------------------------
def embed_y(P, t, masks=None):
    i = 0
    h = t
    while f"{i}.0.weight" in P:
        h = _drop(torch.relu(_lin(P, f"{i}.0", h)), masks, f"y{i}")
        i += 1
    return h


# ---------------------------------------------------------------------------------------
# a10 PrjDiscriminator (RLIP) — model/GANSurv.py:71-105; a11 Discriminator — 52-68
# ---------------------------------------------------------------------------------------
def prj_discriminator(P, x, t, inner_product="instance", prj_path="x", masks=None):
    hid_t = embed_y(_sub(P, "net_pair_two."), t, masks)                 # [1,C']
    hid_x, fc_ins, _ = embed_x_layer(_sub(P, "net_pair_one."), x, masks)
    if inner_product == "bag":
        out = (hid_t * hid_x).sum(dim=-1, keepdim=True)                 # GANSurv.py:92-94
    else:
        out_ins = (fc_ins * hid_t).sum(dim=-1)                          # GANSurv.py:96-97 (B=1 only)
        out = out_ins.mean(dim=-1, keepdim=True)
    if prj_path in ("x", "y"):
        out = out + _lin(P, "prj_layer", hid_x if prj_path == "x" else hid_t)
    return out


def discriminator_cat(P, x, t, masks=None):
    hid_t = embed_y(_sub(P, "net_pair_two."), t, masks)
    hid_x, _, _ = embed_x_layer(_sub(P, "net_pair_one."), x, masks)
    return _lin(P, "fc", torch.cat([hid_x, hid_t], dim=1))


# ---------------------------------------------------------------------------------------
# a12 Generator — model/GANSurv.py:13-49; make_noise_mlp_layer model_utils.py:116-133
# ---------------------------------------------------------------------------------------
def backbone_forward(kind, P, x, x_ext, masks=None):
    if kind == "patch":
        return dualtrans_hs(P, x, masks)
    if kind == "cluster":
        return deep_att_misl(P, x, x_ext, masks)
    if kind == "graph":
        return patch_gcn(P, x.squeeze(0) if x.dim() == 3 else x, x_ext, masks)
    return abmil(P, x, masks)


def generator_head(P, H, noise_flags, noise=None, masks=None, out_scale="sigmoid"):
    """noise: list of tensors, one per layer whose flag is 1 (None/zeros = zero_noise)."""
    nlayers = len(noise_flags)
    it = iter(noise) if noise is not None else None
    for i in range(nlayers):
        if noise_flags[i] == 1:
            n = next(it) if it is not None else torch.zeros_like(H)
            data = torch.cat([H, n], dim=1)                             # GANSurv.py:33-38
        else:
            data = H
        h = _lin(P, f"MLPs.{i}.0", data)
        if i < nlayers - 1:
            if f"MLPs.{i}.1.weight" in P:                               # gen_norm=True variant
                h = F.layer_norm(h, (h.shape[-1],), P[f"MLPs.{i}.1.weight"], P[f"MLPs.{i}.1.bias"], 1e-5)
            h = _drop(torch.relu(h), masks, f"mlp{i}")
        H = h
    if out_scale == "sigmoid":
        return torch.sigmoid(H)
    if out_scale == "exp":
        return torch.exp(H)
    return H


def generator(P, x, x_ext, kind="abmil", noise_flags=(0, 1), noise=None, masks=None,
              out_scale="sigmoid", return_attn=False):
    H, A = backbone_forward(kind, _sub(P, "backbone."), x, x_ext, masks)
    y = generator_head(P, H, list(noise_flags), noise, masks, out_scale)
    return (y, A, H) if return_attn else y


# ---------------------------------------------------------------------------------------
# a13 losses — loss/utils.py
# ---------------------------------------------------------------------------------------
def real_fake_loss(real, fake, which="bce"):
    """loss/utils.py:182-203. The bce fake term is -mean(1 - log(sigmoid(fake)+1e-8)) as shipped."""
    fake = fake.reshape(-1)
    if which == "bce":
        loss = -torch.mean(1.0 - torch.log(torch.sigmoid(fake) + 1e-8))
        if real is not None:
            loss = loss - torch.mean(torch.log(torch.sigmoid(real.reshape(-1)) + 1e-8))
    elif which == "hinge":
        loss = torch.relu(1.0 + fake).mean()
        if real is not None:
            loss = loss + torch.relu(1.0 - real.reshape(-1)).mean()
    elif which == "wasserstein":
        loss = fake.mean()
        if real is not None:
            loss = loss - real.reshape(-1).mean()
    else:
        raise ValueError(which)
    return loss


def fake_generator_loss(fake_score):
    """loss/utils.py:205-208."""
    return -torch.mean(fake_score.reshape(-1))


def recon_loss(pred_t, t, e, alpha=0.0, gamma=1.0, norm="l1"):
    """loss/utils.py:21-41."""
    pred_t, t, e = pred_t.reshape(-1), t.reshape(-1), e.reshape(-1)
    loss_obs = e * torch.abs(pred_t - t)
    loss_cen = (1 - e) * torch.relu(gamma - (pred_t - t))
    if norm == "l2":
        loss_obs = loss_obs * loss_obs
        loss_cen = loss_cen * loss_cen
    loss = (1.0 - alpha) * (loss_obs + loss_cen) + alpha * loss_obs
    return loss.mean()


def loss_reg_l1(coef, params):
    """loss/utils.py:6-14: coef * sum |W| over ALL generator params (biases/LN included)."""
    if coef is None or coef <= 1e-8:
        return 0.0
    return coef * sum(p.abs().sum() for p in params)


# ---------------------------------------------------------------------------------------
# a16 Adam — torch.optim.Adam (L2-in-grad) via optim/optim_factory.py:40-77,
# add_weight_decay 25-37 (no decay on 1-D / *.bias); D: model_handler.py:107
# ---------------------------------------------------------------------------------------
def adam_step(P, G, state, lr, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8, decay_filter=True):
    """In-place on `state` ({'step', 'm', 'v'}); returns the new param dict."""
    b1, b2 = betas
    state["step"] = state.get("step", 0) + 1
    t = state["step"]
    out = {}
    for k, p in P.items():
        g = G.get(k)
        if g is None:
            out[k] = p
            continue
        wd = weight_decay
        if decay_filter and (p.dim() == 1 or k.endswith(".bias")):
            wd = 0.0
        if wd:
            g = g + wd * p
        m = state.setdefault("m", {}).get(k, torch.zeros_like(p))
        v = state.setdefault("v", {}).get(k, torch.zeros_like(p))
        m = b1 * m + (1 - b1) * g
        v = b2 * v + (1 - b2) * g * g
        state["m"][k], state["v"][k] = m, v
        step_size = lr / (1 - b1 ** t)
        denom = v.sqrt() / math.sqrt(1 - b2 ** t) + eps
        out[k] = p - step_size * m / denom
    return out


# ---------------------------------------------------------------------------------------
# a14 the step schedule — model/model_handler.py:349-424 (_update_disc), 426-498 (_update_gen)
# ---------------------------------------------------------------------------------------
class StepConfig:
    """The cfg_nlst.yaml values the step reads."""

    def __init__(self, kind="abmil", noise_flags=(0, 1), out_scale="sigmoid", disc_type="prj",
                 inner_product="instance", prj_path="x", loss_netD="bce", gan_coef=0.004,
                 l1_coef=1e-5, recon_alpha=0.0, recon_gamma=0.0, recon_norm="l1",
                 lr_g=8e-5, wd_g=5e-4, lr_d=8e-5):
        self.__dict__.update(locals())
        del self.__dict__["self"]


def _netD(cfg, PD, x, t, masks):
    if cfg.disc_type == "prj":
        return prj_discriminator(PD, x, t, cfg.inner_product, cfg.prj_path, masks)
    return discriminator_cat(PD, x, t, masks)


def _req(P):
    return {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}


def update_disc(cfg, PG, PD, bags, noise_d, masks_real=None, masks_fake=None,
                n_real_global=None, n_fake_global=None, visible=None):
    """netD.train(), netG.eval() (model_handler.py:355-356). bags = [(x[1,N,C], x_ext, y[1,2])].
    noise_d[i] = generator noise tensors for bag i. masks_real/fake[i] = D dropout masks for the
    real / fake forward of bag i. Returns (losses dict, grads of D, preds, f_fake list).
    n_*_global: denominators when the bags are one rank's shard of a larger step batch."""
    PDg = _req(PD)
    reals, fakes, preds = [], [], []
    for i, (x, x_ext, y) in enumerate(bags):
        t, e = y[:, [0]], y[:, [1]]
        if e.item() == 1 and (visible is None or visible[i]):           # 373-379: event bag with a visible label
            reals.append(_netD(cfg, PDg, x, t, None if masks_real is None else masks_real[i]).reshape(-1))
        with torch.no_grad():                                           # detached at 400
            pred = generator(PG, x, x_ext, cfg.kind, cfg.noise_flags, noise_d[i], None, cfg.out_scale)
        preds.append(pred)
        fakes.append(_netD(cfg, PDg, x, pred, None if masks_fake is None else masks_fake[i]).reshape(-1))
    real = torch.cat(reals) if reals else None
    fake = torch.cat(fakes)
    if n_fake_global is None:
        loss = real_fake_loss(real, fake, cfg.loss_netD)
    else:  # sum-form with global denominators (bag-parallel shard), bce only
        loss = -(1.0 - torch.log(torch.sigmoid(fake) + 1e-8)).sum() / n_fake_global
        if real is not None:
            loss = loss - torch.log(torch.sigmoid(real) + 1e-8).sum() / n_real_global
    loss.backward()
    grads = {k: v.grad for k, v in PDg.items() if v.grad is not None}
    logs = {"Loss_D": loss.item(), "D_real": 0.0 if real is None else real.mean().item(),
            "D_fake": fake.mean().item()}
    return logs, grads, preds, [f.detach() for f in fakes]


def update_gen(cfg, PG, PD, bags, noise_g, masks_g=None, n_global=None, visible=None):
    """netD.eval(), netG.train() (model_handler.py:432-433). visible[i] False = label invisible ('wolabel' mode): the bag
    still feeds the adversarial term but not the supervised loss (473-480)."""
    PGg = _req(PG)
    preds, fakes = [], []
    for i, (x, x_ext, y) in enumerate(bags):
        pred = generator(PGg, x, x_ext, cfg.kind, cfg.noise_flags, noise_g[i],
                         None if masks_g is None else masks_g[i], cfg.out_scale)
        preds.append(pred)
        fakes.append(_netD(cfg, PD, x, pred, None).reshape(-1))
    fake = torch.cat(fakes)
    keep = [i for i in range(len(bags)) if visible is None or visible[i]]
    P_ = torch.cat([preds[i] for i in keep]) if keep else None
    T_ = torch.cat([bags[i][2][:, [0]] for i in keep]) if keep else None
    E_ = torch.cat([bags[i][2][:, [1]] for i in keep]) if keep else None
    if n_global is None:
        gen_loss = fake_generator_loss(fake)                            # 472
        t_reg = (recon_loss(P_, T_, E_, cfg.recon_alpha, cfg.recon_gamma, cfg.recon_norm) if keep
                 else torch.zeros(()))                                  # 479-480
    else:
        gen_loss = -fake.sum() / n_global
        t_reg = recon_loss(P_, T_, E_, cfg.recon_alpha, cfg.recon_gamma, cfg.recon_norm) * (len(bags) / n_global)
    total = t_reg + cfg.gan_coef * gen_loss if cfg.gan_coef != 0.0 else t_reg  # 481-484
    l1 = loss_reg_l1(cfg.l1_coef, PGg.values())                         # 485
    if n_global is None:
        total = total + l1
        total.backward()
    else:  # L1 gradient is added once after the all-reduce; keep it out of the shard's backward
        total.backward()
        total = total + l1
    grads = {k: v.grad for k, v in PGg.items() if v.grad is not None}
    logs = {"Loss_G_fake": gen_loss.item(), "Loss_G_time": t_reg.item(),
            "Loss_G_total": float(total.detach()) if torch.is_tensor(total) else float(total), "D_fake_avg": fake.mean().item()}
    return logs, grads, preds


def train_step(cfg, PG, PD, stG, stD, bags, noise_d, noise_g, masks_real=None, masks_fake=None, masks_g=None, visible=None):
    """One optimizer step of `_train_each_epoch` (model_handler.py:321-345): D update, then
    gen_updates=1 G update against the UPDATED D. Returns (PG', PD', logs, y_hat, f_fake)."""
    logs_d, gD, preds, f_fake = update_disc(cfg, PG, PD, bags, noise_d, masks_real, masks_fake, visible=visible)
    PD2 = adam_step(PD, gD, stD, cfg.lr_d, 0.0, decay_filter=False)
    logs_g, gG, _ = update_gen(cfg, PG, PD2, bags, noise_g, masks_g, visible=visible)
    PG2 = adam_step(PG, gG, stG, cfg.lr_g, cfg.wd_g, decay_filter=True)
    logs = dict(logs_d)
    logs.update(logs_g)
    return PG2, PD2, logs, torch.cat(preds), torch.cat(f_fake), gG, gD


# ---------------------------------------------------------------------------------------
# eval sampling — MyHandler.test_model, model_handler.py:598-643
# ---------------------------------------------------------------------------------------
def test_model_bag(cfg, PG, PD, x, x_ext, noise0, noise_list):
    """Eval mode: y_hat with noise0, f_fake = D(x, y_hat), then len(noise_list) more generator
    samples and their median (torch.median: lower of the two middles)."""
    with torch.no_grad():
        y_hat = generator(PG, x, x_ext, cfg.kind, cfg.noise_flags, noise0, None, cfg.out_scale)
        f_fake = _netD(cfg, PD, x, y_hat, None)
        ys = [generator(PG, x, x_ext, cfg.kind, cfg.noise_flags, n, None, cfg.out_scale) for n in noise_list]
        dist = torch.stack(ys) if ys else None
        avg = torch.median(dist, dim=0)[0] if ys else None
    return y_hat, f_fake, dist, avg


# ---------------------------------------------------------------------------------------
# SURVEY 8f #3 — the supervised baselines: model/BaseSurv.py:22-40 (SurvNet), loss/utils.py:82-181 (MSE_loss, SurvMLE,
# SurvPLE), model/baseline_handler.py:328-368 (_update_network). Pinned by golden G7 (tests/golden/gen_golden.py).
# ---------------------------------------------------------------------------------------
def surv_net(P, x, x_ext, kind="abmil", masks=None, out_scale="sigmoid", hops=1):
    """BaseSurv.py:22-40: backbone -> out_layer (make_noise_mlp_layer with all noise flags 0, then Sigmoid if out_scale ==
    'sigmoid'). out_layer.{i}.0 = Linear; hidden layers are Linear (-> LayerNorm) -> ReLU -> Dropout."""
    H, _ = backbone_forward(kind, _sub(P, "backbone."), x, x_ext, masks)
    n = 1 + hops
    for i in range(n):
        h = _lin(P, f"out_layer.{i}.0", H)
        if i < n - 1:
            if f"out_layer.{i}.1.weight" in P:
                h = F.layer_norm(h, (h.shape[-1],), P[f"out_layer.{i}.1.weight"], P[f"out_layer.{i}.1.bias"], 1e-5)
            h = _drop(torch.relu(h), masks, f"mlp{i}")
        H = h
    return torch.sigmoid(H) if out_scale == "sigmoid" else H


def mse_loss(pred_t, t, e, include_censored=False):
    """loss/utils.py:82-96."""
    pred_t, t, e = pred_t.squeeze(), t.squeeze(), e.squeeze()
    loss = e * (pred_t - t) * (pred_t - t)
    if include_censored:
        loss = loss + (1 - e) * (pred_t - t) * (pred_t - t)
    return loss.mean()


def surv_mle(hazards_hat, t, e, alpha=0.0, eps=1e-7):
    """loss/utils.py:99-135 (SurvMLE.forward): discrete-time NLL; t = bin index, e = event indicator."""
    b = len(t)
    t = t.view(b, 1).long()
    c = 1 - e.view(b, 1).float()
    S = torch.cumprod(1 - hazards_hat, dim=1)
    S_padded = torch.cat([torch.ones_like(c), S], 1)
    unc = -(1 - c) * (torch.log(torch.gather(S_padded, 1, t).clamp(min=eps)) + torch.log(torch.gather(hazards_hat, 1, t).clamp(min=eps)))
    cen = -c * torch.log(torch.gather(S_padded, 1, t + 1).clamp(min=eps))
    neg_l = cen + unc
    return ((1.0 - alpha) * neg_l + alpha * unc).mean()


def surv_ple(y_hat, T, E):
    """loss/utils.py:138-175 (SurvPLE.forward): Breslow partial likelihood; R[i,j] = (T[j] >= T[i]); predictions capped at 10."""
    y_hat = torch.where(y_hat > 10.0, torch.full_like(y_hat, 10.0), y_hat)
    Tf = T.reshape(-1)
    R = (Tf.view(1, -1) >= Tf.view(-1, 1)).float()
    theta = y_hat.reshape(-1)
    # As shipped (loss/utils.py:171-173) the [B] vector is multiplied by E of shape [B,1] (the handler passes label columns,
    # baseline_handler.py:350-352), which broadcasts to [B,B]: the loss is mean_j(theta_j - logsum_j) * mean_i(E_i). Restated as is.
    return -torch.mean((theta - torch.log(torch.sum(torch.exp(theta) * R, dim=1))) * E.float().reshape(-1, 1))


def baseline_step(P, state, bags, kind="abmil", task="surv_reg", out_scale="sigmoid", hops=1, masks=None, lr=8e-5, weight_decay=5e-4,
                  l1_coef=1e-5, recon=(0.0, 0.0, "l1"), mle_alpha=0.0, use_censored=False):
    """baseline_handler.py:328-368: per-bag forwards of the step batch, ONE loss over the concatenated predictions
    (+ L1 over all parameters), backward, Adam. bags = [(x, x_ext, y[1,2])...]. Returns (new P, logs, preds)."""
    Pg = _req(P)
    preds = [surv_net(Pg, x, ext, kind, None if masks is None else masks[i], out_scale, hops) for i, (x, ext, y) in enumerate(bags)]
    cur = torch.cat(preds, dim=0)
    t = torch.cat([b[2][:, [0]] for b in bags], dim=0)
    e = torch.cat([b[2][:, [1]] for b in bags], dim=0)
    if task == "surv_nll":
        net_loss = surv_mle(cur, t, e, mle_alpha)
    elif task == "surv_cox":
        net_loss = surv_ple(cur, t, e)
    elif kind == "patch":
        net_loss = mse_loss(cur, t, e, use_censored)                    # baseline_handler.py:100-104
    else:
        net_loss = recon_loss(cur, t, e, recon[0], recon[1], recon[2])
    total = net_loss + loss_reg_l1(l1_coef, Pg.values())
    total.backward()
    grads = {k: v.grad for k, v in Pg.items() if v.grad is not None}
    newP = adam_step(P, grads, state, lr, weight_decay)
    logs = {"loss_supervision": float(net_loss.detach()), "loss_total": float(total.detach())}
    return newP, logs, [p.detach() for p in preds]
