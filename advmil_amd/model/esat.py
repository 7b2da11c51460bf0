"""ESAT instance self-attention: the post-norm TransformerEncoder(1 layer) that the reference builds with
nn.TransformerEncoderLayer(d_model, nhead, dim_feedforward=d_model, dropout, relu, batch_first)
(model/backbone_utils.py:113-127). Same parameter names (layers.0.self_attn.in_proj_weight, ...) and the
same initialisation: nn.MultiheadAttention / nn.Linear / nn.LayerNorm are kept as parameter holders.

Dense projections (in-proj, out-proj, FFN) run on the HIP GEMM engine; the attention core
(QK^T -> softmax -> dropout -> PV) is the fused flash-style kernel behind ops.mha (csrc/attn.hip); the two post-norm
residuals LayerNorm(x + dropout(.)) are one launch each (ops.add_dropout_layer_norm).
"""
import torch.nn as nn

from .. import ops


class HipTransformerEncoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward, dropout):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout, batch_first=True)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.nhead = nhead

    def forward_rows(self, x2, seg=None):
        """x2[L_total, d]: one bag's region tokens, or a slab of bags (`seg`); attention never crosses a bag boundary,
        everything else in the layer is row-wise and runs once over the slab."""
        rng = getattr(self, "rng", None) or ops.default_rng(x2.device)
        tr = self.training
        sa = self.self_attn
        # [L_total, 3d]; consumed by the attention kernels alone, which read operand planes: the in-projection writes planes ONLY (bf16x3 mode)
        qkv = ops.linear_act(x2, sa.in_proj_weight, sa.in_proj_bias, "none", emit_planes="only")
        # attention core: one fused launch for all (ragged) bags of the slab; `rng_rowoff` (set by the handler under
        # bag-parallel) addresses the dropout row ids of the single-process run
        o = ops.mha(qkv, self.nhead, sa.dropout if tr else 0.0, rng, seg=seg, rowoff=getattr(seg, "rng_rowoff", None))
        o = ops.linear_act(o, sa.out_proj.weight, sa.out_proj.bias, "none")
        x2 = ops.add_dropout_layer_norm(x2, o, self.norm1.weight, self.norm1.bias, self.norm1.eps,
                                        self.dropout1.p if tr else 0.0, rng, "esat_drop1")
        # (every [L, d] activation of the layer reaches its contraction with operand planes: the region embedding and the two LayerNorm
        # outputs leave their kernels with them, the FFN's hidden rows get them from this epilogue, the attention output from one split pass)
        f = ops.linear_act(x2, self.linear1.weight, self.linear1.bias, "relu", self.dropout.p if tr else 0.0, rng, "esat_ffn", emit_planes=True)
        f = ops.linear_act(f, self.linear2.weight, self.linear2.bias, "none")
        return ops.add_dropout_layer_norm(x2, f, self.norm2.weight, self.norm2.bias, self.norm2.eps,
                                          self.dropout2.p if tr else 0.0, rng, "esat_drop2")

    def forward(self, x):
        if x.dim() != 3 or x.shape[0] != 1:
            raise ValueError("ESAT layer: batch_size 1 expected")
        return self.forward_rows(x[0]).unsqueeze(0)


class HipTransformerEncoder(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward, dropout, num_layers):
        super().__init__()
        self.layers = nn.ModuleList([HipTransformerEncoderLayer(d_model, nhead, dim_feedforward, dropout)
                                     for _ in range(num_layers)])

    def forward_rows(self, x2, seg=None):
        for layer in self.layers:
            x2 = layer.forward_rows(x2, seg)
        return x2

    def forward(self, x):
        for layer in self.layers:
            x = layer(x)
        return x
