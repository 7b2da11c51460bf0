"""MIL backbones behind the reference's `load_backbone(mode, dims)` (model/backbone.py:19-51):
ABMIL (54-86), DeepAttMISL (89-123), DualTrans_HS = ESAT (171-196), PatchGCN (126-168).
Each is `forward(x, x_ext, *args) -> H[1, dim_out]` with the reference's state_dict keys; the N-row
work is in the HIP library (advmil_amd/ops.py)."""
from types import SimpleNamespace
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..utils.func import dropout_small
from .backbone_utils import Attn_Net_Gated, GAPool, make_embedding_layer, make_transformer_layer, _rng_of


def Model_Zoo(mode):
    return {"patch": DualTrans_HS, "cluster": DeepAttMISL, "graph": PatchGCN}.get(mode, ABMIL)


def load_backbone_param(mode, dims):
    if mode == "patch":
        emb = SimpleNamespace(in_dim=dims[0], out_dim=dims[1], scale=4, dw_conv=False, ksize=1)
        tra = SimpleNamespace(d_model=dims[1], nhead=8, dropout=0.25, num_layers=1)
        return [dims[:3], "avgpool", emb, "Transformer", tra], {"dropout": 0.25}
    if mode == "cluster":
        return [dims[:3]], {"num_clusters": 8, "dropout": 0.25}
    if mode == "graph":
        return [dims[:3]], {"num_layers": 1, "dropout": 0.25}
    return [dims[:3]], {"dropout": 0.25}


def load_backbone(mode, dims):
    args, kws = load_backbone_param(mode, dims)
    return Model_Zoo(mode)(*args, **kws)


def _small_fc(seq, x, rng, tag):
    """Linear -> ReLU -> Dropout on a [1, d] vector (launch-bound; stays a couple of tiny device ops)."""
    lin, drop = seq[0], seq[2]
    return ops.linear_act_any(x, lin.weight, lin.bias, "relu", drop.p if seq.training else 0.0, rng, tag)   # one launch, dropout in the epilogue


class ABMIL(nn.Module):
    def __init__(self, dims: List, dropout: float = 0.25):
        super().__init__()
        assert len(dims) == 3
        dim_in, dim_hid, dim_out = dims
        self.attention_net = nn.Sequential(nn.Linear(dim_in, dim_hid), nn.ReLU(), nn.Dropout(dropout),
                                           Attn_Net_Gated(L=dim_hid, D=dim_hid, dropout=dropout, n_classes=1))
        self.rho = nn.Sequential(nn.Linear(dim_hid, dim_out), nn.ReLU(), nn.Dropout(dropout))

    def features_multi(self, X, seg=None, exts=None):
        """The N-row part over a slab X[N_total, C] of B bags (`seg`; None = one bag): -> pooled [B, hid]
        (everything before `rho`). One FC GEMM, one gate GEMM and one segmented softmax-pool for the whole step batch."""
        rng = _rng_of(self, X)
        fc, p = self.attention_net[0], (self.attention_net[2].p if self.training else 0.0)
        # the epilogue also emits h's operand planes: the gate contraction below reads them through the plane-fed kernel
        gate = self.attention_net[3]
        pg = float(getattr(gate, "drop_p", 0.0)) if gate.training else 0.0
        # (gate_sites: the pool below draws the scorer's two dropout sites next -- their keep bits ride in this layer's dropout launch)
        # (no-grad passes -- evaluation, the eval forward of a D update that did not take the two-layer launch: planes INSTEAD of fp32 rows)
        only = (not torch.is_grad_enabled()) and p <= 0.0 and ops.H_PLANES_ONLY and fc.weight.shape[0] % 32 == 0
        h = ops.linear_act(X, fc.weight, fc.bias, "relu", p, rng, "abmil_fc", emit_planes="only" if only else True,
                           gate_sites=(pg, "gate_att_a", "gate_att_b") if (pg > 0.0 and _rng_of(gate, X) is rng) else None)   # [N_total, hid]
        pooled, A, _ = self.attention_net[3].pool(h, seg)
        self.last_attention = A.detach()
        return pooled.unsqueeze(0) if seg is None else pooled

    def features(self, x_path, *args):
        return self.features_multi(x_path.squeeze(0))            # batch_size = 1

    def post(self, pooled):
        """`rho` on a [B, hid] stack of pooled bags (B = the bags of one optimizer step)."""
        return _small_fc(self.rho, pooled, _rng_of(self, pooled), "abmil_rho")

    def forward(self, x_path, *args):
        return self.post(self.features(x_path, *args))


class DeepAttMISL(nn.Module):
    def __init__(self, dims: List, num_clusters=8, dropout=0.25):
        super().__init__()
        assert len(dims) == 3
        dim_in, dim_hid, dim_out = dims
        assert dim_hid == dim_out
        self.dim_hid, self.num_clusters = dim_hid, num_clusters
        self.phis = nn.Sequential(nn.Conv2d(dim_in, dim_hid, 1), nn.ReLU())
        self.pool1d = nn.AdaptiveAvgPool1d(1)
        self.attention_net = nn.Sequential(nn.Linear(dim_hid, dim_hid), nn.ReLU(), nn.Dropout(dropout),
                                           Attn_Net_Gated(L=dim_hid, D=dim_hid, dropout=dropout, n_classes=1))

    def features_multi(self, X, seg=None, exts=None):
        """Slab form. exts = per-bag cluster-id tensors. Per-patch FC+ReLU (the 1x1 conv), then the per-(bag, cluster)
        means as ONE [8B, N_total] x [N_total, hid] contraction with the normalised membership matrix: no host sync on the
        ids, no per-cluster gathers, empty cluster -> zeros (reference: D2H of ids + python loop of boolean gathers,
        backbone.py:107-116); then the 8-row attention pool per bag."""
        rng = _rng_of(self, X)
        K = self.num_clusters
        h = ops.linear_act(X, self.phis[0].weight, self.phis[0].bias, "relu")
        cid = torch.cat([e.reshape(-1) for e in exts]).to(device=X.device, dtype=torch.long)
        nb = 1 if seg is None else seg.nseg
        if seg is not None:
            cid = cid + seg.rowseg.to(torch.long) * K
        h_cluster = ops.segmented_mean(h, cid, K * nb)                            # [8B, hid]
        fc, p = self.attention_net[0], (self.attention_net[2].p if self.training else 0.0)
        hc = ops.linear_act_any(h_cluster, fc.weight, fc.bias, "relu", p, rng, "misl_fc")
        pooled, A, _ = self.attention_net[3].pool(hc, None if seg is None else seg.uniform(K))
        self.last_attention = A.detach()
        return pooled.unsqueeze(0) if seg is None else pooled

    def forward(self, x_path, cluster_id, *args):
        return self.features_multi(x_path.squeeze(0), None, [cluster_id])


class DualTrans_HS(nn.Module):
    """ESAT: region embedding (FC+LN+ReLU+mean16) -> 1 post-norm transformer layer -> GAPool."""

    def __init__(self, dims: List, emb_backbone: str, args_emb_backbone, tra_backbone: str, args_tra_backbone,
                 dropout: float = 0.25):
        super().__init__()
        assert len(dims) == 3
        dim_in, dim_hid, dim_out = dims
        assert dim_hid == dim_out
        assert emb_backbone in ["avgpool", "gapool"]
        assert tra_backbone in ["Transformer", "Identity"]
        self.patch_embedding_layer = make_embedding_layer(emb_backbone, args_emb_backbone)
        self.dim_hid = dim_hid
        self.patch_encoder_layer = make_transformer_layer(tra_backbone, args_tra_backbone)
        self.pool = GAPool(dim_out, dim_out)

    def features_multi(self, X, seg=None, exts=None):
        """Slab form: region embedding and every projection / FFN of the transformer layer run once over all bags'
        rows; only the attention core and the final GAPool respect bag boundaries."""
        seg16 = None if seg is None else seg.div(16)
        emb = self.patch_embedding_layer.embed_rows(X)
        enc = self.patch_encoder_layer
        feat = enc.forward_rows(emb, seg16) if hasattr(enc, "forward_rows") else emb
        H = self.pool.pool_rows(feat, seg16)
        self.last_attention = self.pool.last_attention
        return H

    def forward(self, x, coord=None, *args):
        if coord is not None:
            raise NotImplementedError("positional encoding is unreachable in the reference (model_handler.py:390 passes None)")
        return self.features_multi(x.squeeze(0))


class _GENConvParams(nn.Module):
    """Parameter holder with torch_geometric.nn.GENConv's state_dict keys: `t` [1] and `mlp.{0,1,4}`
    (Linear(d,2d), LayerNorm(2d), ReLU, Dropout, Linear(2d,d); num_layers=2, norm='layer')."""

    def __init__(self, dim, t=1.0):
        super().__init__()
        self.t = nn.Parameter(torch.tensor([float(t)]))
        self.mlp = nn.Sequential(nn.Linear(dim, 2 * dim), nn.LayerNorm(2 * dim), nn.ReLU(), nn.Dropout(0.0), nn.Linear(2 * dim, dim))
        self.eps = 1e-7


class _DeepGCNLayerParams(nn.Module):
    """DeepGCNLayer(conv, norm, act): only `conv` runs when num_layers == 1 (model/backbone.py:157); `norm` exists in
    the reference's state_dict, so it exists here."""

    def __init__(self, dim):
        super().__init__()
        self.conv = _GENConvParams(dim)
        self.norm = nn.LayerNorm(dim, elementwise_affine=True)
        self.act = nn.ReLU(inplace=True)


class PatchGCN(nn.Module):
    """FC -> GENConv(softmax aggr, learnable t) -> cat -> FC -> gated attention pool. `x_path` is a graph object with
    `.x [N, dim_in]` and `.edge_index [2, E]` (a torch_geometric Batch in the reference; any object with those two
    attributes here). Parity unpinned (GENConv's arithmetic is torch_geometric's, absent and unversioned)."""

    def __init__(self, dims: List, num_layers: int = 3, edge_agg: str = "spatial", dropout: float = 0.25):
        super().__init__()
        assert len(dims) == 3
        dim_in, dim_hid, dim_out = dims
        if num_layers != 1:
            raise NotImplementedError("HIP PatchGCN covers num_layers=1, the value load_backbone_param ships (backbone.py:40)")
        self.edge_agg, self.num_layers = edge_agg, num_layers
        self.fc = nn.Sequential(nn.Linear(dim_in, dim_hid), nn.ReLU(), nn.Dropout(dropout))
        self.layers = nn.ModuleList([_DeepGCNLayerParams(dim_hid) for _ in range(num_layers)])
        self.path_phi = nn.Sequential(nn.Linear(dim_hid * (1 + num_layers), dim_out), nn.ReLU(), nn.Dropout(dropout))
        self.path_attention_head = Attn_Net_Gated(L=dim_out, D=dim_out, dropout=dropout, n_classes=1)

    def features_multi(self, X, seg=None, exts=None):
        """Slab form: the bags' graphs become one block-diagonal graph over the slab's rows (node ids offset per bag)."""
        if seg is None:
            return self.forward(exts[0])
        # one union CSR per distinct group of graphs, kept alive for the module's lifetime: captured HIP graphs hold the
        # raw addresses of these index arrays
        key = tuple(id(e) for e in exts)
        caches = self.__dict__.setdefault("_union_cache", {})
        cache = caches.get(key)
        if cache is None:
            ei = torch.cat([e.edge_index.to(torch.long) + seg.offsets[b] for b, e in enumerate(exts)], dim=1)
            from types import SimpleNamespace
            union = SimpleNamespace(x=X, edge_index=ei)
            union._advmil_csr = ops.GraphCSR(ei, seg.total)
            caches[key] = cache = (key, union, list(exts))
        union = cache[1]
        union.x = X
        return self._run(union, seg)

    def forward(self, x_path, *args):
        return self._run(x_path, None)

    def _run(self, data, seg):
        x_in = data.x
        rng = _rng_of(self, x_in)
        tr = self.training
        csr = ops.graph_csr(data)
        x = ops.linear_act(x_in, self.fc[0].weight, self.fc[0].bias, "relu", self.fc[2].p if tr else 0.0, rng, "gcn_fc")
        conv = self.layers[0].conv
        agg = ops.genconv_aggregate(x, conv.t, csr, conv.eps)
        hmid = ops.linear_act(agg, conv.mlp[0].weight, conv.mlp[0].bias, "none")
        hmid = ops.ln_relu(hmid, conv.mlp[1].weight, conv.mlp[1].bias, conv.mlp[1].eps)
        x1 = ops.linear_act(hmid, conv.mlp[4].weight, conv.mlp[4].bias, "none")
        h = torch.cat([x, x1], dim=1)
        h = ops.linear_act(h, self.path_phi[0].weight, self.path_phi[0].bias, "relu", self.path_phi[2].p if tr else 0.0, rng, "gcn_phi")
        pooled, A, _ = self.path_attention_head.pool(h, seg)
        self.last_attention = A.detach()
        return pooled.unsqueeze(0) if seg is None else pooled
