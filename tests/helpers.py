"""Shared test helpers: synthetic params/bags as torch tensors, oracle-side shapes."""
import numpy as np
import torch

from advmil_amd import synth

DATA_SEED, PARAM_SEED = 0, 42


def T(a, device="cpu"):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


# state_dict shapes of the reference modules (probed) --------------------
def shapes_attn_gated(prefix, d):
    return {f"{prefix}attention_a.0.weight": (d, d), f"{prefix}attention_a.0.bias": (d,),
            f"{prefix}attention_b.0.weight": (d, d), f"{prefix}attention_b.0.bias": (d,),
            f"{prefix}attention_c.weight": (1, d), f"{prefix}attention_c.bias": (1,)}


def shapes_gapool(prefix, d):
    return {f"{prefix}fc1.0.weight": (d, d), f"{prefix}fc1.0.bias": (d,),
            f"{prefix}score.0.weight": (d, d), f"{prefix}score.0.bias": (d,),
            f"{prefix}fc2.weight": (1, d), f"{prefix}fc2.bias": (1,)}


def shapes_head(d=384, out=1):
    # make_noise_mlp_layer(384, 1, noise=[0,1], hops=1): MLPs.0 = Linear(384,192)+ReLU+Drop, MLPs.1 = Linear(384,1)
    return {"MLPs.0.0.weight": (d // 2, d), "MLPs.0.0.bias": (d // 2,),
            "MLPs.1.0.weight": (out, d), "MLPs.1.0.bias": (out,)}


def shapes_generator(kind, c=1024, d=384):
    s = dict(shapes_head(d))
    b = "backbone."
    if kind == "abmil":
        s.update({b + "attention_net.0.weight": (d, c), b + "attention_net.0.bias": (d,),
                  b + "rho.0.weight": (d, d), b + "rho.0.bias": (d,)})
        s.update(shapes_attn_gated(b + "attention_net.3.", d))
    elif kind == "cluster":
        s.update({b + "phis.0.weight": (d, c, 1, 1), b + "phis.0.bias": (d,),
                  b + "attention_net.0.weight": (d, d), b + "attention_net.0.bias": (d,)})
        s.update(shapes_attn_gated(b + "attention_net.3.", d))
    elif kind == "patch":
        e, t = b + "patch_embedding_layer.", b + "patch_encoder_layer.layers.0."
        s.update({e + "conv.weight": (d, c, 1, 1), e + "conv.bias": (d,), e + "norm.weight": (d,), e + "norm.bias": (d,),
                  t + "self_attn.in_proj_weight": (3 * d, d), t + "self_attn.in_proj_bias": (3 * d,),
                  t + "self_attn.out_proj.weight": (d, d), t + "self_attn.out_proj.bias": (d,),
                  t + "linear1.weight": (d, d), t + "linear1.bias": (d,), t + "linear2.weight": (d, d), t + "linear2.bias": (d,),
                  t + "norm1.weight": (d,), t + "norm1.bias": (d,), t + "norm2.weight": (d,), t + "norm2.bias": (d,)})
        s.update(shapes_gapool(b + "pool.", d))
    elif kind == "graph":
        d = 128
        s = dict(shapes_head(d))
        c_ = b + "layers.0.conv."
        s.update({b + "fc.0.weight": (d, c), b + "fc.0.bias": (d,), c_ + "t": (1,),
                  c_ + "mlp.0.weight": (2 * d, d), c_ + "mlp.0.bias": (2 * d,), c_ + "mlp.1.weight": (2 * d,), c_ + "mlp.1.bias": (2 * d,),
                  c_ + "mlp.4.weight": (d, 2 * d), c_ + "mlp.4.bias": (d,),
                  b + "layers.0.norm.weight": (d,), b + "layers.0.norm.bias": (d,),
                  b + "path_phi.0.weight": (d, 2 * d), b + "path_phi.0.bias": (d,)})
        s.update(shapes_attn_gated(b + "path_attention_head.", d))
    else:
        raise ValueError(kind)
    return s


def shapes_disc(disc_type="prj", prj_path="x", c=1024, d=128):
    p = "net_pair_one."
    s = {p + "embedding.conv.weight": (d, c, 1, 1), p + "embedding.conv.bias": (d,),
         p + "embedding.norm.weight": (d,), p + "embedding.norm.bias": (d,),
         p + "fc1.0.weight": (d // 2, d), p + "fc1.0.bias": (d // 2,), p + "fc1.3.weight": (d, d // 2), p + "fc1.3.bias": (d,),
         p + "fc2.0.weight": (d // 2, d), p + "fc2.0.bias": (d // 2,), p + "fc2.3.weight": (d, d // 2), p + "fc2.3.bias": (d,),
         "net_pair_two.0.0.weight": (64, 1), "net_pair_two.0.0.bias": (64,),
         "net_pair_two.1.0.weight": (d, 64), "net_pair_two.1.0.bias": (d,)}
    s.update(shapes_gapool(p + "pool.", d))
    if disc_type == "prj":
        if prj_path in ("x", "y"):
            s.update({"prj_layer.weight": (1, d), "prj_layer.bias": (1,)})
    else:
        s.update({"fc.weight": (1, 2 * d), "fc.bias": (1,)})
    return s


def synth_params(shapes, prefix, seed=PARAM_SEED, device="cpu"):
    """Same naming as tests/golden/gen_golden.py::load_synth (prefix + state_dict key)."""
    return {k: T(synth.param(seed, prefix + k, s), device) for k, s in shapes.items()}


def noise_tensor(tag, k, width, device="cpu"):
    return T(synth.device_uniform(DATA_SEED, synth.stream_key(7, f"{tag}:{k}"), width).reshape(1, width), device)


def bag(i, n, device="cpu"):
    return T(synth.bag(DATA_SEED, i, n), device)


def label(i, device="cpu"):
    return T(synth.label(DATA_SEED, i), device)


def poison_host_bag(x):
    """NaN everywhere in a host bag except the five elements the bag cache samples as the bag's identity
    (advmil_amd.ingest.bag_fingerprint): a later epoch that still READ the host bag would produce NaNs, while the cache keeps
    recognising it as the same bag. Returns the poisoned tensor (a new one)."""
    import torch
    from advmil_amd.ingest import fingerprint_positions
    y = torch.full_like(x, float("nan"))
    fx, fy = x.reshape(-1), y.reshape(-1)
    for i in fingerprint_positions(fx.numel()):
        fy[i] = fx[i]
    return y
