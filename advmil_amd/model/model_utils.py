"""model/model_utils.py of the reference: init_weights (12-17), general_init_weight / init_pytorch_defaults (20-71),
get_hop_dims / make_noise_mlp_layer (106-133),
make_efficient_mlp_layer (157-166), make_mlp_layer (168-176), make_embedding_y_layer (178-186),
EmbedXLayer (188-210). Same names, arguments and state_dict keys."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..utils.func import dropout_small
from .backbone_utils import GAPool, make_embedding_layer, _rng_of


@torch.no_grad()
def init_weights(m):
    if isinstance(m, nn.Linear):
        nn.init.xavier_uniform_(m.weight)
        if m.bias is not None:
            m.bias.data.zero_()


@torch.no_grad()
def general_init_weight(m):
    """model_utils.py:20-71 with version='041' (the PyTorch 0.4.1 defaults, halved for Linear): used by the Cox baseline."""
    import math
    if isinstance(m, nn.Linear):
        stdv = 1.0 / math.sqrt(m.weight.size(1)) * 0.5
        m.weight.data.uniform_(-stdv, stdv)
        if m.bias is not None:
            m.bias.data.uniform_(-stdv, stdv)
    elif isinstance(m, nn.Conv2d):
        n = m.in_channels
        for k in m.kernel_size:
            n *= k
        stdv = 1.0 / math.sqrt(n)
        m.weight.data.uniform_(-stdv, stdv)
        if m.bias is not None:
            m.bias.data.uniform_(-stdv, stdv)


def get_hop_dims(d, hops):
    dims, cur = [], d
    for _ in range(hops):
        cur //= 2
        if cur <= 1:
            break
        dims.append(cur)
    return dims


LN_RELU_MAX_WIDTH = 512     # advmil_ln_relu_fwd / _bwd (csrc/pool.hip): one row per wave, the row held in registers


def make_mlp_layer(dim_in, dim_out, layer_norm=True, dropout=0.25):
    layers = [nn.Linear(dim_in, dim_out), nn.ReLU(inplace=True), nn.Dropout(dropout)]
    if layer_norm:
        if dim_out > LN_RELU_MAX_WIDTH:
            # (fail where the model is BUILT, naming the width, not with an EINVAL from the first step)
            raise ValueError(f"advmil_amd: a LayerNorm MLP layer of width {dim_out} has no HIP path (the LayerNorm -> ReLU row kernel takes "
                             f"<= {LN_RELU_MAX_WIDTH} columns): lower the hid_dims / pdh_dims entry that asks for it, or set norm: false")
        layers.insert(1, nn.LayerNorm(dim_out))
    return nn.Sequential(*layers)


def make_noise_mlp_layer(in_dim: int, out_dim: int, noise, hops: int = 1, norm: bool = False, dropout: float = 0.25):
    hid = get_hop_dims(in_dim, hops)
    ins, outs = [in_dim] + hid, hid + [out_dim]
    mlps = nn.ModuleList()
    for i, (di, do) in enumerate(zip(ins, outs)):
        di = di * 2 if noise[i] == 1 else di          # noise is concatenated at full width (GANSurv.py:33-38)
        if i == len(outs) - 1:
            mlps.append(nn.Sequential(nn.Linear(di, do)))
        else:
            mlps.append(make_mlp_layer(di, do, norm, dropout))
    return mlps


def make_efficient_mlp_layer(dim, layer_norm=True, dropout=0.25):
    if layer_norm:
        raise NotImplementedError("the reference raises NameError here (model_utils.py:165); it is only called with False")
    return nn.Sequential(nn.Linear(dim, dim // 2), nn.ReLU(inplace=True), nn.Dropout(dropout), nn.Linear(dim // 2, dim))


def make_embedding_y_layer(args):
    layers, d = [], args.in_dim
    for h in args.hid_dims:
        layers.append(make_mlp_layer(d, h, args.norm, args.dropout))
        d = h
    return nn.Sequential(*layers)


def _fusable(lin, x):
    """The HIP contraction engine takes the layer when every contiguous extent is a multiple of 4 floats."""
    return x.dim() == 2 and x.is_cuda and lin.in_features % 4 == 0 and lin.out_features % 4 == 0


def run_mlp_small(seq, x, rng, tag, final_act=None):
    """Apply a Sequential of Linear / LayerNorm / ReLU / Dropout holders to a [B, d]-sized tensor. Linear (+ReLU) (+Dropout)
    runs of a layer go through ONE contraction launch (bias, activation and dropout in its epilogue; in the backward the weight
    and bias gradients are accumulated straight into the optimizer's arena), the rest stays a tiny device op each.
    final_act ('sigmoid'): an activation the CALLER applies to the result (the generator's out_scale); when the Sequential ends in a
    width-1 Linear it rides in that layer's launch. Then returns (x, applied)."""
    mods = list(seq)
    applied = False
    j = 0
    while j < len(mods):
        m = mods[j]
        if isinstance(m, nn.Sequential):
            x = run_mlp_small(m, x, rng, f"{tag}.{j}")
        elif isinstance(m, nn.Linear):
            if _fusable(m, x):
                act, p, k = "none", 0.0, j + 1
                if k < len(mods) and isinstance(mods[k], nn.ReLU):
                    act, k = "relu", k + 1
                    if k < len(mods) and isinstance(mods[k], nn.Dropout):
                        p, k = (mods[k].p if seq.training else 0.0), k + 1
                x = ops.linear_act(x, m.weight, m.bias, act, p, rng, f"{tag}.{k - 1}")   # the Dropout module's index names the draw
                j = k
                continue
            if x.dim() == 2 and x.is_cuda and (m.in_features == 1 or m.out_features == 1) and x.shape[0] <= 256:
                # in/out width 1 (first layer of the y-embedding, output layers): one launch each way; a following ReLU rides along
                act, k = "none", j + 1
                if k < len(mods) and isinstance(mods[k], nn.ReLU):
                    act, k = "relu", k + 1
                elif k == len(mods) and final_act is not None:
                    act, applied = final_act, True
                x = ops.skinny_linear(x, m.weight, m.bias, act)
                j = k
                continue
            # any other width (not a multiple of 4 floats): zero-padded through the same contraction (ops.linear_act_any)
            act, p, k = "none", 0.0, j + 1
            if k < len(mods) and isinstance(mods[k], nn.ReLU):
                act, k = "relu", k + 1
                if k < len(mods) and isinstance(mods[k], nn.Dropout):
                    p, k = (mods[k].p if seq.training else 0.0), k + 1
            x = ops.linear_act_any(x, m.weight, m.bias, act, p, rng, f"{tag}.{k - 1}")
            j = k
            continue
        elif isinstance(m, nn.LayerNorm):
            # make_mlp_layer(layer_norm=True): Linear -> LayerNorm -> ReLU (-> Dropout): LayerNorm + ReLU is one row kernel each way
            if not (j + 1 < len(mods) and isinstance(mods[j + 1], nn.ReLU) and len(m.normalized_shape) == 1 and x.dim() == 2):
                raise NotImplementedError("advmil_amd: a LayerNorm that is not followed by ReLU has no HIP path (the reference's builders never make one)")
            x = ops.ln_relu(x.contiguous(), m.weight, m.bias, m.eps)
            j += 2
            continue
        elif isinstance(m, nn.ReLU):
            x = F.relu(x)
        elif isinstance(m, nn.Sigmoid):
            x = torch.sigmoid(x)
        elif isinstance(m, nn.Dropout):
            x = dropout_small(x, m.p, seq.training, rng, f"{tag}.{j}")
        else:
            raise NotImplementedError(type(m))
        j += 1
    return x if final_act is None else (x, applied)


class EmbedXLayer(nn.Module):
    """[1, N, C] -> region embedding [1, N/16, C'] -> per-region MLP -> GAPool -> MLP -> [1, C']."""

    def __init__(self, args):
        super().__init__()
        out_dim = args.out_dim
        args.scale = 4
        args.dw_conv = False
        self.embedding = make_embedding_layer(args.backbone, args)
        self.fc1 = make_efficient_mlp_layer(out_dim, False, args.dropout)
        self.pool = GAPool(out_dim, out_dim, args.dropout)
        self.fc2 = make_efficient_mlp_layer(out_dim, False, args.dropout)

    def embed(self, x):
        """The dropout-free, t-independent part: x -> emb_ins[1, L, C'] (shared by the real and fake pairs)."""
        return self.embedding(x)

    def pool_features_rows(self, e, seg16=None, want_mean=False):
        """Region-level part over rows e[L_total, C'] (one bag or a slab): -> (emb_bag[B,C'] pooled before fc2, fc_ins[L_total,C']);
        want_mean: a third result, the per-bag mean of fc_ins [B, C'] (the RLIP inner product's operand). fc1 and the pooling's scorer
        run as ONE launch each way when the widths and the arithmetic mode allow it (ops.dx_region_pool, csrc/region.hip)."""
        rng = _rng_of(self, e)
        tr = self.training
        f0, f3, pl = self.fc1[0], self.fc1[3], self.pool
        p1 = self.fc1[2].p if tr else 0.0
        if e.dim() == 2 and ops.dx_chain_ok(e, f0.weight, f0.bias, f3.weight, f3.bias, pl.fc1[0].weight, pl.fc1[0].bias, pl.score[0].weight,
                                            pl.score[0].bias, pl.fc2.weight, pl.fc2.bias):
            pooled, mean, A, fc_ins = ops.dx_region_pool(e, f0.weight, f0.bias, f3.weight, f3.bias, pl.fc1[0].weight, pl.fc1[0].bias,
                                                         pl.score[0].weight, pl.score[0].bias, pl.fc2.weight, pl.fc2.bias, p1,
                                                         pl.drop_p if pl.training else 0.0, rng, seg16, want_mean)
            pl.last_attention = A.detach()
            if seg16 is None:
                pooled = pooled.reshape(1, -1)
                mean = None if mean is None else mean.reshape(1, -1)
            return (pooled, fc_ins, mean) if want_mean else (pooled, fc_ins)
        h = ops.linear_act(e, f0.weight, f0.bias, "relu", p1, rng, "dx_fc1")
        fc_ins = ops.linear_act(h, f3.weight, f3.bias, "none")
        pooled = self.pool.pool_rows(fc_ins, seg16)
        if want_mean:
            return pooled, fc_ins, (ops.segmented_mean_rows(fc_ins, seg16) if seg16 is not None else fc_ins.mean(dim=0, keepdim=True))
        return pooled, fc_ins

    def pool_features(self, emb_ins):
        """emb_ins[1,L,C'] -> (emb_bag[1,C'], fc_ins[1,L,C'])."""
        emb_bag, fc_ins = self.pool_features_rows(emb_ins[0])
        return emb_bag, fc_ins.unsqueeze(0)

    def from_embedding(self, emb_ins, return_instance=False):
        emb_bag, fc_ins = self.pool_features(emb_ins)
        fc_bag = run_mlp_small(self.fc2, emb_bag, _rng_of(self, emb_ins), "dx_fc2")
        return (fc_bag, fc_ins) if return_instance else fc_bag

    def forward(self, x, return_instance=False):
        return self.from_embedding(self.embed(x), return_instance)
