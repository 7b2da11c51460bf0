"""Layers of the reference's model/backbone_utils.py with the same class names, ctor arguments and
state_dict keys; every forward over N (patches) or N/16 (regions) rows runs in the HIP library.

  Attn_Net_Gated          model/backbone_utils.py:11-29
  GAPool                  model/backbone_utils.py:31-56
  sequence2square/...     model/backbone_utils.py:62-77   (folded into the fused embedding tail)
  make_embedding_layer    model/backbone_utils.py:101-111
  make_transformer_layer  model/backbone_utils.py:113-127
  AVGPoolPatchEmbedding   model/backbone_utils.py:129-168

The nn.Linear / nn.Conv2d / nn.LayerNorm children are parameter holders (identical keys and init);
forward never calls them.
"""
import torch
import torch.nn as nn

from .. import ops


def _rng_of(module, x):
    return getattr(module, "rng", None) or ops.default_rng(x.device)


class Attn_Net_Gated(nn.Module):
    """Gated attention scorer: A = c(tanh(a x) * sigmoid(b x)). `dropout` truthy -> Dropout(0.25) on both
    branches, as in the reference (backbone_utils.py:17-19)."""

    def __init__(self, L=1024, D=256, dropout=False, n_classes=1):
        super().__init__()
        a = [nn.Linear(L, D), nn.Tanh()]
        b = [nn.Linear(L, D), nn.Sigmoid()]
        if dropout:
            a.append(nn.Dropout(0.25))
            b.append(nn.Dropout(0.25))
        self.attention_a = nn.Sequential(*a)
        self.attention_b = nn.Sequential(*b)
        self.attention_c = nn.Linear(D, n_classes)
        self.drop_p = 0.25 if dropout else 0.0
        if n_classes != 1 or L != D:
            raise NotImplementedError("HIP gated-attention kernels cover n_classes=1 and L==D (all AdvMIL configs)")

    def pool(self, x, seg=None):
        """Fused scorer + softmax over instances + weighted sum: x[N,D] -> (pooled[D], A[N], raw scores[N]).
        With `seg` (ops.Segments) the rows are a slab of B bags and pooled is [B, D] (softmax per bag)."""
        p = self.drop_p if self.training else 0.0
        return ops.gated_attn_pool(x, self.attention_a[0].weight, self.attention_a[0].bias, self.attention_b[0].weight,
                                   self.attention_b[0].bias, self.attention_c.weight, self.attention_c.bias, p, _rng_of(self, x), "gate_", seg)

    def forward(self, x):
        """Reference contract: (A[N,1] raw scores, x)."""
        p = self.drop_p if self.training else 0.0
        s = ops.gate_scores(x, self.attention_a[0].weight, self.attention_a[0].bias, self.attention_b[0].weight,
                            self.attention_b[0].bias, self.attention_c.weight, self.attention_c.bias, p, _rng_of(self, x))
        return s.reshape(-1, 1), x


class GAPool(nn.Module):
    """Global attention pooling of a [1, L, d] sequence -> [1, d]; pools its input."""

    def __init__(self, in_dim, hid_dim, dropout=0.25):
        super().__init__()
        self.fc1 = nn.Sequential(nn.Linear(in_dim, hid_dim), nn.Tanh(), nn.Dropout(dropout))
        self.score = nn.Sequential(nn.Linear(in_dim, hid_dim), nn.Sigmoid(), nn.Dropout(dropout))
        self.fc2 = nn.Linear(hid_dim, 1)
        self.drop_p = float(dropout)
        if in_dim != hid_dim:
            raise NotImplementedError("HIP GAPool covers in_dim == hid_dim (all AdvMIL configs)")

    def pool_rows(self, x2, seg=None):
        """x2[L_total, d] (one bag, or a slab of bags partitioned by `seg`) -> pooled [B, d]."""
        p = self.drop_p if self.training else 0.0
        pooled, A, _ = ops.gated_attn_pool(x2, self.fc1[0].weight, self.fc1[0].bias, self.score[0].weight,
                                           self.score[0].bias, self.fc2.weight, self.fc2.bias, p, _rng_of(self, x2), "gapool_", seg)
        self.last_attention = A.detach()
        return pooled.unsqueeze(0) if seg is None else pooled

    def forward(self, x):
        if x.dim() != 3 or x.shape[0] != 1:
            raise ValueError("GAPool: the AdvMIL path is batch_size 1 (config/cfg_nlst.yaml:70); got %s" % (tuple(x.shape),))
        return self.pool_rows(x[0])


class AVGPoolPatchEmbedding(nn.Module):
    """FC (1x1 conv on 4x4 tiles == row-wise FC) -> LayerNorm -> ReLU -> mean over each consecutive 16 patches.
    [1, N, C] -> [1, N/16, out_dim]; N % 16 == 0."""

    def __init__(self, in_dim, out_dim, scale: int = 4, dw_conv=False, ksize=3, stride=1):
        super().__init__()
        assert scale == 4, "It only supports for scale = 4"
        assert ksize == 1 or ksize == 3, "It only supports for ksize = 1 or 3"
        if ksize != 1 or stride != 1 or dw_conv:
            raise NotImplementedError("HIP embedding covers ksize=1, stride=1 (cfg disc_netx_ksize: 1; ESAT ksize 1)")
        self.scale, self.stride = scale, stride
        self.conv = nn.Conv2d(in_dim, out_dim, ksize, stride, padding=(ksize - 1) // 2)
        self.pool = nn.AdaptiveAvgPool2d(1)
        self.norm = nn.LayerNorm(out_dim)
        self.act = nn.ReLU(inplace=True)

    def embed_rows(self, x2, dup=1):
        """x2[N_total, C] -> [N_total/16, out_dim]; rows may be a slab of bags (each bag a multiple of 16 rows, so the
        16-row regions never straddle two bags). dup = 2: the embedding twice, stacked ([emb; emb], the discriminator update's fake |
        real batch) straight from the kernel -- its backward sums the two halves of the gradient on load."""
        assert x2.shape[0] % (self.scale * self.scale) == 0
        if dup != 1 and not ops.ln_relu_mean16_dup_ok(self.norm.normalized_shape[0]):
            e = self.embed_rows(x2)
            return torch.cat([e] * dup, dim=0)
        b = self.conv.bias
        gb = ops._arena_grad(b) if (b is not None and b.requires_grad and torch.is_grad_enabled()) else None
        if gb is not None:
            # the FC's bias gradient is the column sum of dy, which the LayerNorm backward produces while it writes dy: the FC
            # itself sees a constant bias and never re-reads dy (805 MB at the 32768-patch slab) for it
            y = ops.linear_act(x2, self.conv.weight, b.detach(), "none")
            return ops.ln_relu_mean16(y, self.norm.weight, self.norm.bias, self.norm.eps, ycol_grad=gb.view(-1), dup=dup)
        y = ops.linear_act(x2, self.conv.weight, b, "none")
        return ops.ln_relu_mean16(y, self.norm.weight, self.norm.bias, self.norm.eps, dup=dup)

    def forward(self, x):
        if x.dim() != 3 or x.shape[0] != 1:
            raise ValueError("AVGPoolPatchEmbedding: batch_size 1 expected, got %s" % (tuple(x.shape),))
        return self.embed_rows(x[0]).unsqueeze(0)


def make_embedding_layer(backbone: str, args):
    if backbone == "avgpool":
        return AVGPoolPatchEmbedding(args.in_dim, args.out_dim, args.scale, args.dw_conv, args.ksize)
    raise NotImplementedError(f"{backbone}: only `avgpool` is reachable from the shipped configs (cfg_nlst.yaml:42)")


def make_transformer_layer(backbone: str, args):
    if backbone == "Transformer":
        from .esat import HipTransformerEncoder
        return HipTransformerEncoder(args.d_model, args.nhead, args.d_model, args.dropout, args.num_layers)
    if backbone == "Identity":
        return nn.Identity()
    raise NotImplementedError(f"{backbone} has not implemented.")
