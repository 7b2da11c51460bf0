"""Generator / Discriminator / PrjDiscriminator (RLIP) of the reference's model/GANSurv.py (13-49, 52-68,
71-105): same ctor arguments, forward signatures and state_dict keys."""
import torch
import torch.nn as nn

from .. import ops
from ..utils.func import generate_noise
from .backbone_utils import _rng_of
from .model_utils import EmbedXLayer, make_embedding_y_layer, make_noise_mlp_layer, run_mlp_small


class Generator(nn.Module):
    def __init__(self, dim_in, dim_out, backbone: nn.Module, args_noise, norm=False, dropout=0.25, out_scale: str = "sigmoid"):
        super().__init__()
        self.noise = args_noise.noise
        self.hops = args_noise.hops
        self.noise_dist = "uniform" if args_noise.noise_dist is None else args_noise.noise_dist
        assert len(self.noise) == self.hops + 1
        self.MLPs = make_noise_mlp_layer(dim_in, dim_out, self.noise, hops=self.hops, norm=norm, dropout=dropout)
        self.backbone = backbone
        self.out_scale = out_scale

    def head(self, H, zero_noise=False, noise=None):
        """H[1, dim_in] -> prediction. `noise`: optional list of injected tensors (one per noisy layer)."""
        rng = _rng_of(self, H)
        it = iter(noise) if noise is not None else None
        for i, layer in enumerate(self.MLPs):
            if self.noise[i] == 1:
                if zero_noise:
                    n = torch.zeros_like(H)
                elif it is not None:
                    n = next(it).to(H.device)
                else:
                    n = generate_noise(*H.size(), to_device=H.device, distribution=self.noise_dist, rng=rng)
                H = torch.cat([H, n], dim=1)
            if i == len(self.MLPs) - 1 and self.out_scale == "sigmoid":      # the output scale rides in the last layer's launch
                H, done = run_mlp_small(layer, H, rng, f"gen_mlp{i}", final_act="sigmoid")
                if done:
                    return H
            else:
                H = run_mlp_small(layer, H, rng, f"gen_mlp{i}")
        if self.out_scale == "sigmoid":
            return torch.sigmoid(H)
        if self.out_scale == "exp":
            return torch.exp(H)
        return H

    def features(self, x, x_ext):
        """Per-bag N-row work only; the [1,d]-sized remainder (`finish`) can then run once for a whole step batch."""
        bb = self.backbone
        return bb.features(x, x_ext) if hasattr(bb, "features") else bb(x, x_ext)

    def features_multi(self, X, seg, exts=None):
        """Slab form of `features`: X[N_total, C] holding the B bags of a step back to back -> [B, d]."""
        return self.backbone.features_multi(X, seg, exts)

    def _head_spec(self, feats, zero_noise, noise):
        """ops.GHeadSpec when `finish` can run as the fused launches (csrc/ghead.hip): the shipped head -- an optional backbone `rho`
        (Linear -> ReLU -> Dropout), MLPs[0] = Linear -> ReLU -> Dropout, uniform / zero / injected noise or none, a width-1 output layer,
        out_scale sigmoid or none -- on <= 32 bags. Decided BEFORE any call site is drawn (the layer-by-layer path draws its own); the
        sites are then drawn in that path's order and under its tags (abmil_rho, gen_mlp0.2, noise)."""
        if not (ops.GHEAD and feats.is_cuda and feats.dim() == 2 and feats.dtype == torch.float32 and feats.shape[0] <= 32):
            return None
        if len(self.MLPs) != 2 or list(self.noise) not in ([0, 1], [0, 0]) or self.out_scale == "exp":
            return None
        bb = self.backbone
        rho = None
        if hasattr(bb, "post"):
            rho = getattr(bb, "rho", None)
            if not (isinstance(rho, nn.Sequential) and len(rho) == 3 and isinstance(rho[0], nn.Linear) and isinstance(rho[1], nn.ReLU)
                    and isinstance(rho[2], nn.Dropout)):
                return None
        m0, m1 = self.MLPs[0], self.MLPs[1]
        if not (isinstance(m0, nn.Sequential) and len(m0) == 3 and isinstance(m0[0], nn.Linear) and isinstance(m0[1], nn.ReLU)
                and isinstance(m0[2], nn.Dropout) and isinstance(m1, nn.Sequential) and len(m1) == 1 and isinstance(m1[0], nn.Linear)
                and m1[0].out_features == 1):
            return None
        B, d2 = feats.shape[0], m0[0].out_features
        has_noise = self.noise[1] == 1
        if not has_noise:
            mode, nz = 0, None
        elif zero_noise:
            mode, nz = 1, None
        elif noise is not None:
            if len(noise) != 1:
                return None
            nz = noise[0].to(feats.device, torch.float32).contiguous()
            if tuple(nz.shape) != (B, d2):
                return None
            mode = 2
        elif self.noise_dist == "uniform":
            mode, nz = 3, None
        else:
            return None
        p1 = (rho[2].p if rho.training else 0.0) if rho is not None else 0.0
        p2 = m0[2].p if m0.training else 0.0
        probe = ops.GHeadSpec(Wr=None if rho is None else rho[0].weight, br=None if rho is None else rho[0].bias, p1=p1, W0=m0[0].weight,
                              b0=m0[0].bias, p2=p2, W1=m1[0].weight, b1=m1[0].bias, noise_mode=mode, noise=nz,
                              out_act=1 if self.out_scale == "sigmoid" else 0)
        if not ops.ghead_ok(feats, probe):
            return None
        rng = _rng_of(self, feats)
        # the fused kernels take ONE row map for all their draws: every active site's map must be the same object (looked up BEFORE any site
        # is drawn, so that a mismatch can still fall back to the layer-by-layer path, which draws the sites itself)
        tags = ((["abmil_rho"] if p1 > 0.0 else []) + (["gen_mlp0.2"] if p2 > 0.0 else []) + (["noise"] if mode == 3 else []))
        ok, rr = _one_row_map(rng, B, tags)
        if not ok:
            return None
        probe.rr = rr
        if p1 > 0.0:
            probe.sid1, probe.seed = rng.site("abmil_rho", (B, rho[0].out_features), p1), rng.seed
        if p2 > 0.0:
            probe.sid2, probe.seed = rng.site("gen_mlp0.2", (B, d2), p2), rng.seed
        if mode == 3:
            probe.sid_noise, probe.seed = rng.site("noise", (B * d2,), None), rng.seed
        return probe

    def finish(self, feats, zero_noise=False, noise=None, pred_out=None):
        """feats[B, d] (stacked `features`) -> predictions [B, dim_out]. pred_out: an fp32 [B, 1] buffer the predictions are written into
        when the fused head runs without a graph (the D update's stacked label column); the caller checks data_ptr()."""
        spec = self._head_spec(feats, zero_noise, noise)
        if spec is not None:                 # rho + head as two launches each way
            return ops.ghead(feats, spec, pred_out if not torch.is_grad_enabled() else None)
        bb = self.backbone
        H = bb.post(feats) if hasattr(bb, "post") else feats
        return self.head(H, zero_noise, noise)

    def forward(self, x, x_ext, zero_noise=False, noise=None):
        return self.finish(self.features(x, x_ext), zero_noise, noise)


def _one_row_map(rng, n_rows, tags):
    """(True, the row map shared by the call sites `tags` for a tensor of n_rows rows) -- or (False, None) when two of them resolve to
    different maps (ops.SITE_LAYOUTS): a fused kernel that draws several sites from one map would then differ from the layer-by-layer
    and the single-process draws without any error."""
    rr, first = None, True
    for t in tags:
        m = rng.row_map(n_rows, t)
        if first:
            rr, first = m, False
        elif m is not rr:                    # (identity, not torch.equal: no device sync inside a step that may be under graph capture)
            return False, None
    return True, rr


class _PairNet(nn.Module):
    def __init__(self, args_netx, args_nety):
        super().__init__()
        self.net_pair_one = EmbedXLayer(args_netx)
        self.net_pair_two = make_embedding_y_layer(args_nety)

    def embed_x(self, x):
        return self.net_pair_one.embed(x)

    def embed_rows(self, X, dup=1):
        """Slab form: X[N_total, C] -> region embeddings [N_total/16, C'] (dup = 2: stacked twice)."""
        return self.net_pair_one.embedding.embed_rows(X, dup) if dup != 1 else self.net_pair_one.embedding.embed_rows(X)

    def forward(self, x, t):
        return self.from_embedding(self.embed_x(x), t)


class Discriminator(_PairNet):
    """Concat fusion (disc_type: cat)."""

    def __init__(self, args_netx, args_nety, **kws):
        super().__init__(args_netx, args_nety)
        self.fc = nn.Linear(args_netx.out_dim + args_nety.hid_dims[-1], 1)

    def bag_features(self, emb_ins):
        """Per-bag region-level work (fc1 MLP + GAPool): (emb_bag[1,C'], None)."""
        emb_bag, _ = self.net_pair_one.pool_features(emb_ins)
        return emb_bag, None

    def bag_features_multi(self, emb, seg16):
        emb_bag, _ = self.net_pair_one.pool_features_rows(emb, seg16)
        return emb_bag, None

    def tail(self, emb_bag, ins_mean, t):
        """[B,C'] stacks + t[B,1] -> f[B,1]: fc2, net_pair_two and the concat head, once per step batch."""
        rng = _rng_of(self, t)
        hid_x = run_mlp_small(self.net_pair_one.fc2, emb_bag, rng, "dx_fc2")
        hid_t = run_mlp_small(self.net_pair_two, t, rng, "dy")
        return ops.linear_act_any(torch.cat([hid_x, hid_t], dim=1).contiguous(), self.fc.weight, self.fc.bias)   # [B, 1]: padded to the 4-float granularity

    def from_embedding(self, emb_ins, t):
        return self.tail(*self.bag_features(emb_ins), t)


class PrjDiscriminator(_PairNet):
    """Projection discriminator; inner_product='instance' is RLIP (region-level inner product)."""

    def __init__(self, args_netx, args_nety, prj_path="x", inner_product="bag"):
        super().__init__(args_netx, args_nety)
        assert inner_product in ["bag", "instance"]
        self.inner_product = inner_product
        dim_x, dim_y = args_netx.out_dim, args_nety.hid_dims[-1]
        self.prj_path = prj_path
        if prj_path == "x":
            self.prj_layer = nn.Linear(dim_x, 1)
        elif prj_path == "y":
            self.prj_layer = nn.Linear(dim_y, 1)
        else:
            self.prj_layer = None

    def bag_features(self, emb_ins):
        """Per-bag region-level work that does not depend on t (fc1 MLP + GAPool, HIP kernels):
        (emb_bag[1,C'], mean_r fc_ins[1,C'] | None). RLIP is linear in the region mean:
        mean_r(fc_ins_r . hid_t) == mean_r(fc_ins_r) . hid_t (GANSurv.py:96-98)."""
        emb_bag, fc_ins = self.net_pair_one.pool_features(emb_ins)
        return emb_bag, (fc_ins.mean(dim=1) if self.inner_product == "instance" else None)

    def bag_features_multi(self, emb, seg16):
        """Slab form of `bag_features`: emb[L_total, C'] + region segments -> (emb_bag[B,C'], mean_r fc_ins[B,C'] | None)."""
        if self.inner_product != "instance":
            return self.net_pair_one.pool_features_rows(emb, seg16)[0], None
        emb_bag, _, mean = self.net_pair_one.pool_features_rows(emb, seg16, want_mean=True)
        return emb_bag, mean

    def tail(self, emb_bag, ins_mean, t):
        """[B,C'] stacks + t[B,1] -> f[B,1]: fc2, net_pair_two, the (region-level) inner product and the projection,
        once per step batch instead of once per bag."""
        rng = _rng_of(self, t)
        spec = self._tail_spec(emb_bag, ins_mean, t, rng)
        if spec is not None:                 # the whole tail as ONE launch each way (csrc/tail.hip)
            return ops.dtail(emb_bag, ins_mean if self.inner_product == "instance" else None, t, spec)
        hid_x = run_mlp_small(self.net_pair_one.fc2, emb_bag, rng, "dx_fc2")
        hid_t = run_mlp_small(self.net_pair_two, t, rng, "dy")
        if hid_x.shape[0] <= 256 and hid_x.is_cuda and hid_x.dtype == torch.float32:      # [B, d] head: one launch each way
            u = hid_x if self.inner_product == "bag" else ins_mean
            if self.prj_layer is None:
                return ops.prj_head(u, hid_t)
            return ops.prj_head(u, hid_t, hid_x if self.prj_path == "x" else hid_t, self.prj_layer.weight, self.prj_layer.bias)
        if self.inner_product == "bag":
            out = (hid_t * hid_x).sum(dim=-1, keepdim=True)
        else:
            out = (ins_mean * hid_t).sum(dim=-1, keepdim=True)
        if self.prj_layer is not None:
            src = hid_x if self.prj_path == "x" else hid_t
            out = out + (ops.skinny_linear(src, self.prj_layer.weight, self.prj_layer.bias) if src.shape[0] <= 256
                         else ops.linear_act_any(src.contiguous(), self.prj_layer.weight, self.prj_layer.bias))
        return out

    def _tail_spec(self, emb_bag, ins_mean, t, rng):
        """ops.TailSpec of this tail when it can run as the fused launch (plain Linear -> ReLU -> Dropout chains, <= 32 rows), else None.
        The dropout call sites are drawn in the order and under the tags of the layer-by-layer path (dx_fc2.*, dy.*)."""
        if not (ops.DTAIL and emb_bag.is_cuda and emb_bag.dim() == 2 and emb_bag.shape[0] <= 32 and emb_bag.dtype == torch.float32
                and t.dim() == 2 and t.dtype == torch.float32):
            return None
        if self.inner_product == "instance" and ins_mean is None:
            return None
        chains = []
        for seq, tag, tr in ((self.net_pair_one.fc2, "dx_fc2", self.net_pair_one.fc2.training), (self.net_pair_two, "dy", self.net_pair_two.training)):
            flat = []                        # (module, tag of a Dropout) with nested Sequentials opened, tags as run_mlp_small forms them
            def walk(s, tg):
                for j, m in enumerate(s):
                    if isinstance(m, nn.Sequential):
                        walk(m, f"{tg}.{j}")
                    else:
                        flat.append((m, f"{tg}.{j}"))
            walk(seq, tag)
            layers, j = [], 0
            while j < len(flat):
                m = flat[j][0]
                if not isinstance(m, nn.Linear):
                    return None
                act, p, ptag, k = 0, 0.0, None, j + 1
                if k < len(flat) and isinstance(flat[k][0], nn.ReLU):
                    act, k = 1, k + 1
                if k < len(flat) and isinstance(flat[k][0], nn.Dropout):
                    p, ptag, k = (flat[k][0].p if tr else 0.0), flat[k][1], k + 1
                layers.append((m, act, float(p), ptag))
                j = k
            chains.append(layers)
        B = emb_bag.shape[0]
        if chains[0][0][0].in_features != emb_bag.shape[1] or chains[1][0][0].in_features != t.shape[1]:
            return None
        prj_src = 0 if self.prj_layer is None else (1 if self.prj_path == "x" else 2)
        prj_w = None if self.prj_layer is None else self.prj_layer.weight
        prj_b = None if self.prj_layer is None else self.prj_layer.bias
        probe = ops.TailSpec(*[[(m.weight, m.bias, act, p, 0) for (m, act, p, _) in layers] for layers in chains], prj_w, prj_b, prj_src, None, None)
        if not ops.dtail_ok(B, probe):       # (decided BEFORE any call site is drawn: the layer-by-layer path draws its own)
            return None
        ok, rr = _one_row_map(rng, B, [ptag for layers in chains for (_, _, p, ptag) in layers if p > 0.0])
        if not ok:                           # (sites whose maps differ: the layer-by-layer path, one map per site)
            return None
        spec_l, seed = [], None
        for layers in chains:
            out = []
            for (m, act, p, ptag) in layers:
                sid = 0
                if p > 0.0:
                    sid, seed = rng.site(ptag, (B, m.out_features), p), rng.seed
                out.append((m.weight, m.bias, act, p, sid))
            spec_l.append(out)
        return ops.TailSpec(spec_l[0], spec_l[1], prj_w, prj_b, prj_src, seed, rr)

    def from_embedding(self, emb_ins, t):
        return self.tail(*self.bag_features(emb_ins), t)
