"""float64 pre-activations of the generator's ReLU layers that lie within fp32 round-off of the boundary (shared by the fuzz tools): such an
entry takes either ReLU branch in ANY fp32 evaluation depending on the summation order -- tiles, split-K, the slab pad, the rank count --; the
forward does not move (relu(+-1e-7) = 0 to fp32), that unit's weight / bias gradient row and everything upstream of it does, and Adam turns the
difference into steps of up to lr per entry. A property of the input, not of a kernel."""
import torch


def generator_boundary_entries(kind, PG, bags, band=2e-6):
    """bags: [(x [1, N, 1024], cluster ids | None, label)]; PG: the generator's parameters by name. -> list of (level, where..., value)."""
    P = {k: v.detach().double().cpu() for k, v in PG.items()}
    out = []
    first = {"abmil": "backbone.attention_net.0", "cluster": "backbone.phis.0"}.get(kind)
    if first is None:
        return out
    W1 = P[first + ".weight"].reshape(P[first + ".weight"].shape[0], -1)
    b1 = P[first + ".bias"]
    row0 = 0
    for bi_, (x_, e_, _) in enumerate(bags):
        z = x_.reshape(-1, x_.shape[-1]).double().cpu() @ W1.t() + b1
        for r, c in (z.abs() < band).nonzero().tolist():
            out.append(("patch", row0 + r, c, float(z[r, c])))
        if kind == "cluster":
            W2, b2 = P["backbone.attention_net.0.weight"], P["backbone.attention_net.0.bias"]
            h = torch.relu(z)
            ids = e_.reshape(-1).long().cpu()
            m = torch.stack([h[ids == c].mean(dim=0) if bool((ids == c).any()) else torch.zeros(h.shape[1], dtype=torch.float64) for c in range(8)])
            z2 = m @ W2.t() + b2
            for r, c in (z2.abs() < band).nonzero().tolist():
                out.append(("cluster", bi_, r, c, float(z2[r, c])))
        row0 += z.shape[0]
    return out
