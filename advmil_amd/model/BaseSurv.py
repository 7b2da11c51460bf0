"""model/BaseSurv.py of the reference (SurvNet, 22-40): backbone + prediction head without noise, the network of the supervised
baselines (NLL / Cox / regression). Same ctor arguments and state_dict keys (`backbone.*`, `out_layer.{i}.{j}.*`)."""
import torch.nn as nn

from .backbone_utils import _rng_of
from .model_utils import make_noise_mlp_layer, run_mlp_small


class SurvNet(nn.Module):
    def __init__(self, dim_in, dim_out, backbone: nn.Module, hops=1, norm=False, dropout=0.25, out_scale="none"):
        super().__init__()
        self.backbone = backbone
        noise = [0] * (1 + hops)             # no noise in the forward; same head builder as the Generator
        mlps = make_noise_mlp_layer(dim_in, dim_out, noise, hops=hops, norm=norm, dropout=dropout)
        if out_scale == "sigmoid":
            self.out_layer = nn.Sequential(*mlps, nn.Sigmoid())
        elif out_scale == "none":
            self.out_layer = nn.Sequential(*mlps)
        else:
            raise ValueError(f"out_scale={out_scale}: the reference builds no head for it (BaseSurv.py:33-34)")

    def features(self, x, x_ext):
        bb = self.backbone
        return bb.features(x, x_ext) if hasattr(bb, "features") else bb(x, x_ext)

    def features_multi(self, X, seg, exts=None):
        """Slab form: X[N_total, C] holding the B bags of a step back to back -> [B, d] (the N-row kernels run once)."""
        return self.backbone.features_multi(X, seg, exts)

    def finish(self, feats):
        """feats[B, d] -> predictions [B, dim_out]: the backbone's [B,d]-sized tail and the head, once per step batch."""
        bb = self.backbone
        H = bb.post(feats) if hasattr(bb, "post") else feats
        return run_mlp_small(self.out_layer, H, _rng_of(self, H), "surv_mlp")

    def forward(self, x, x_ext):
        return self.finish(self.features(x, x_ext))
