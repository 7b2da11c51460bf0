"""Default configuration of the G+D path: the keys of the reference's config/cfg_nlst.yaml that the step reads
(model_handler.py:37-137, 301-498), with that file's values. `bcb_mode` defaults to the ABMIL generator used by
BASELINE.json configs 1-3; 'patch' selects ESAT (config 4), 'cluster' DeepAttMISL."""


def default_cfg(**over):
    cfg = dict(
        task="cont_gansurv", seed=42, cuda_id=0, save_path=None, test=False,
        bcb_mode="abmil", bcb_dims="1024-384-384",
        gen_dims="384-1", gen_noi_noise="0-1", gen_noi_noise_dist="uniform", gen_noi_hops=1, gen_norm=False,
        gen_dropout=0.6, gen_out_scale="sigmoid",
        disc_type="prj", disc_netx_in_dim=1024, disc_netx_out_dim=128, disc_netx_ksize=1, disc_netx_backbone="avgpool",
        disc_netx_dropout=0.25, disc_nety_in_dim=1, disc_nety_hid_dims="64-128", disc_nety_norm=False, disc_nety_dropout=0.0,
        disc_prj_path="x", disc_prj_iprd="instance",
        loss_gan_coef=0.004, loss_netD="bce", loss_regl1_coef=0.00001, loss_mle_alpha=0.0,
        loss_recon_norm="l1", loss_recon_alpha=0.0, loss_recon_gamma=0.0,
        opt_netG="adam", opt_netG_lr=0.00008, opt_netG_weight_decay=0.0005, opt_netD_lr=0.00008,
        batch_size=1, bp_every_batch=16, gen_updates=1, times_test_sample=30, test_zero_noise=True,
    )
    cfg.update(over)
    return cfg


def default_baseline_cfg(**over):
    """Keys BaselineHandler reads (model/baseline_handler.py:34-120); the reference ships no baseline yaml, the values follow
    cfg_nlst.yaml where the key has a counterpart there."""
    cfg = dict(
        task="surv_reg", seed=42, cuda_id=0, save_path=None, test=False,
        bcb_mode="abmil", bcb_dims="1024-384-384", pdh_dims="384-1", mlp_hops=1, mlp_norm=False, mlp_dropout=0.6,
        loss_regl1_coef=0.00001, loss_mle_alpha=0.0, loss_recon_norm="l1", loss_recon_alpha=0.0, loss_recon_gamma=0.0,
        loss_use_censored=False, time_bins=4,
        opt_net="adam", opt_net_lr=0.00008, opt_net_weight_decay=0.0005,
        batch_size=1, bp_every_batch=16,
    )
    cfg.update(over)
    return cfg
