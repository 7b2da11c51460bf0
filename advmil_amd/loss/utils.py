"""Losses of the G+D step (reference loss/utils.py): loss_reg_l1 (6-14), recon_loss (21-41),
real_fake_loss (182-203), fake_generator_loss (205-208); and of the supervised baselines: MSE_loss (82-96), SurvMLE (99-135),
SurvPLE (138-175). They act on <= bp_every_batch scalars, so
they are a handful of tiny device ops; the one N-sized piece -- sum|W| over the generator arena -- is
the abs_sum kernel, and its gradient is folded into the fused Adam kernel (advmil_amd/optim.py)."""
import torch
import torch.nn.functional as F

from .. import ops


def loss_reg_l1(coef):
    coef = 0.0 if coef is None else coef

    def func(model_params):
        if coef <= 1e-8:
            return 0.0
        params = list(model_params)
        return coef * sum(ops.abs_sum(w.detach().reshape(-1))[0] if w.is_cuda and w.is_contiguous() else w.abs().sum()
                          for w in params)
    return func


def recon_terms(pred_t, t, e, alpha=0.0, gamma=1.0, norm="l1", cur_alpha=None):
    """Per-sample terms whose mean is recon_loss (loss/utils.py:21-41)."""
    pred_t, t, e = pred_t.reshape(-1), t.reshape(-1), e.reshape(-1)
    loss_obs = e * torch.abs(pred_t - t)
    loss_cen = (1 - e) * F.relu(gamma - (pred_t - t))
    if norm == "l2":
        loss_obs, loss_cen = loss_obs * loss_obs, loss_cen * loss_cen
    a = alpha if cur_alpha is None else cur_alpha
    return (1.0 - a) * (loss_obs + loss_cen) + a * loss_obs


def recon_loss(pred_t, t, e, alpha=0.0, gamma=1.0, norm="l1", cur_alpha=None):
    return recon_terms(pred_t, t, e, alpha, gamma, norm, cur_alpha).mean()


def real_fake_terms(real, fake, which="bce"):
    """Per-sample terms whose means make up real_fake_loss: (terms_real | None, terms_fake).
    bce keeps the reference's shipped form -(1 - log(sigmoid(fake)+1e-8)) (loss/utils.py:185-186)."""
    fake = fake.reshape(-1)
    real = None if real is None else real.reshape(-1)
    if which == "bce":
        tf = -(1.0 - torch.log(torch.sigmoid(fake) + 1e-8))
        tr = None if real is None else -torch.log(torch.sigmoid(real) + 1e-8)
    elif which == "hinge":
        tf = F.relu(1.0 + fake)
        tr = None if real is None else F.relu(1.0 - real)
    elif which == "wasserstein":
        tf = fake
        tr = None if real is None else -real
    else:
        raise ValueError(which)
    return tr, tf


def real_fake_loss(real, fake, which="bce"):
    tr, tf = real_fake_terms(real, fake, which)
    loss = tf.mean()
    if tr is not None:
        loss = loss + tr.mean()
    return loss


def fake_generator_loss(fake_score):
    return -torch.mean(fake_score.reshape(-1))


# ---- supervised baselines (model/baseline_handler.py:92-106) -------------------------------------------------------------
def MSE_loss(pred_t, t, e, include_censored=False):
    """loss/utils.py:82-96 (the ESAT baseline's loss)."""
    pred_t, t, e = pred_t.squeeze(), t.squeeze(), e.squeeze()
    loss = e * (pred_t - t) * (pred_t - t)
    if include_censored:
        loss = loss + (1 - e) * (pred_t - t) * (pred_t - t)
    return loss.mean()


class SurvMLE(torch.nn.Module):
    """Discrete-time negative log-likelihood, loss/utils.py:99-135. t = bin index, e = event indicator."""

    def __init__(self, alpha=0.0, eps=1e-7):
        super().__init__()
        self.alpha, self.eps = alpha, eps

    def forward(self, hazards_hat, t, e, cur_alpha=None):
        b = len(t)
        t = t.view(b, 1).long()
        c = 1 - e.view(b, 1).float()
        S = torch.cumprod(1 - hazards_hat, dim=1)
        S_padded = torch.cat([torch.ones_like(c), S], 1)
        unc = -(1 - c) * (torch.log(torch.gather(S_padded, 1, t).clamp(min=self.eps))
                          + torch.log(torch.gather(hazards_hat, 1, t).clamp(min=self.eps)))
        cen = -c * torch.log(torch.gather(S_padded, 1, t + 1).clamp(min=self.eps))
        alpha = self.alpha if cur_alpha is None else cur_alpha
        return ((1.0 - alpha) * (cen + unc) + alpha * unc).mean()


class SurvPLE(torch.nn.Module):
    """Breslow partial likelihood, loss/utils.py:138-175. The risk-set matrix R[i,j] = (T[j] >= T[i]) is one broadcast compare on
    the device (the reference fills it with a python double loop of .item()-style reads). As shipped, the [B] vector of log partial
    likelihoods is multiplied by E of shape [B,1] (the handler passes label columns), which broadcasts to [B,B] before the mean;
    that behaviour is kept."""

    def forward(self, y_hat, T, E):
        y_hat = torch.where(y_hat > 10.0, torch.full_like(y_hat, 10.0), y_hat)
        Tf = T.reshape(-1)
        R = (Tf.view(1, -1) >= Tf.view(-1, 1)).to(y_hat.dtype)
        theta = y_hat.reshape(-1)
        return -torch.mean((theta - torch.log(torch.sum(torch.exp(theta) * R, dim=1))) * E.float())
