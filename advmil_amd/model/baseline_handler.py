"""model/baseline_handler.py of the reference (BaselineHandler: ctor 34-141, _train_each_epoch 287-326, _update_network 328-368,
test_model 443-487, save/resume 489-516) for the three supervised tasks (surv_reg / surv_nll / surv_cox) on the same HIP path as
the G+D handler: the bags of an optimizer step run as ONE slab through the backbone kernels, the [B,d]-sized head and the loss run
once on the stacked predictions, the L1 sub-gradient is folded into the fused Adam launch. Orchestration (exec / k-fold / wandb /
evaluator wiring) stays with the reference."""
import os
import os.path as osp
from functools import partial
from types import SimpleNamespace

import torch

from .. import ops
from ..loss.utils import MSE_loss, SurvMLE, SurvPLE, recon_loss
from ..optim import create_optimizer
from ..utils.func import agg_tensor, seed_everything, sparse_key, sparse_str
from .backbone import load_backbone
from .BaseSurv import SurvNet
from .model_utils import general_init_weight, init_weights


class BaselineHandler(object):
    def __init__(self, cfg, device=None):
        assert cfg["task"] in ["surv_cox", "surv_nll", "surv_reg"]
        assert cfg["bcb_mode"] in ["patch", "cluster", "graph", "abmil"]
        if device is None:
            if torch.cuda.device_count() == 0:
                raise RuntimeError("advmil_amd.BaselineHandler needs an MI355X: no ROCm device visible (no CPU fallback)")
            device = torch.device("cuda", int(cfg.get("cuda_id", 0)) % torch.cuda.device_count())
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        seed_everything(cfg["seed"])
        self.rng = ops.default_rng(self.device)
        self.cfg, self.bcb, self.task = cfg, cfg["bcb_mode"], cfg["task"]
        save_path = cfg.get("save_path")
        if save_path:
            os.makedirs(save_path, exist_ok=True)
            self.last_net_ckpt_path = osp.join(save_path, "model-last.pth")
            self.best_net_ckpt_path = osp.join(save_path, "model-best.pth")
        # out_scale / time format follow the task (baseline_handler.py:68-78)
        out_scale = "none" if self.task == "surv_cox" else "sigmoid"
        cfg["time_format"] = {"surv_nll": "quantile", "surv_reg": "ratio", "surv_cox": "origin"}[self.task]
        backbone = load_backbone(self.bcb, sparse_str(cfg["bcb_dims"]))
        dim_in, dim_out = sparse_str(cfg["pdh_dims"])
        self.net = SurvNet(dim_in, dim_out, backbone, hops=cfg["mlp_hops"], norm=cfg["mlp_norm"], dropout=cfg["mlp_dropout"],
                           out_scale=out_scale)
        self.net.apply(init_weights if self.task in ("surv_reg", "surv_nll") else general_init_weight)       # 87-90
        self.net = self.net.to(self.device)
        # 'fp32' (default) | 'bf16': how bags are held in HBM (MyHandler.x_storage); the static test_model reads it off the network
        self.x_storage = os.environ.get("ADVMIL_X_STORAGE", cfg.get("x_storage", "fp32"))
        self.net._advmil_x_storage = self.x_storage
        for m in self.net.modules():
            m.rng = self.rng
        if cfg.get("gemm_mode") is not None:
            ops.set_gemm_mode(cfg["gemm_mode"])
        if self.task == "surv_nll":                                                                           # 93-105
            self.supervised_loss = SurvMLE(**sparse_key(cfg, prefixes="loss_mle"))
        elif self.task == "surv_cox":
            self.supervised_loss = SurvPLE()
        elif self.bcb == "patch":
            self.supervised_loss = partial(MSE_loss, include_censored=cfg["loss_use_censored"])
        else:
            self.supervised_loss = partial(recon_loss, **sparse_key(cfg, prefixes="loss_recon"))
        self.coef_l1 = 0.0 if cfg["loss_regl1_coef"] is None else float(cfg["loss_regl1_coef"])
        opt_cfg = SimpleNamespace(opt=cfg["opt_net"], weight_decay=cfg["opt_net_weight_decay"], lr=cfg["opt_net_lr"], opt_eps=None,
                                  opt_betas=None, momentum=None)
        self.optimizer = create_optimizer(opt_cfg, self.net)
        self.optimizer.l1_coef = self.coef_l1 if self.coef_l1 > 1e-8 else 0.0        # d/dW coef*sum|W| applied inside the Adam kernel
        self.steplr = torch.optim.lr_scheduler.ReduceLROnPlateau(self.optimizer, mode="min", factor=0.5, patience=10)
        self.patient_id = dict()
        self.history = []

    # ------------------------------------------------------------------------------------------
    def log(self, d):
        self.history.append(d)

    def pop_logs(self):
        out = [{k: (float(v) if torch.is_tensor(v) else v) for k, v in d.items()} for d in self.history]
        self.history = []
        return out

    def _train_each_epoch(self, train_loader, name_loader):
        """Reference contract (287-326): returns {'y', 'y_hat'} collected over the epoch. Host bags go through the pinned staging
        slab on the copy stream and stay in the device-resident bag cache across epochs (advmil_amd/ingest.py::step_batches: the
        same ingest as MyHandler's); the trailing partial step batch is dropped as the reference drops it."""
        from ..ingest import loader_cache_view, new_scope_token, step_batches
        bp = self.cfg["bp_every_batch"]
        ys_all, yh_all = [], []
        gb = self.cfg.get("bag_cache_gb")
        view = None
        if self.bcb != "graph" and (gb is None or float(gb) > 0):
            # (a loader without a dataset object is cached -- under this handler's scope -- only on explicit request: bag_cache_gb set)
            explicit = gb is not None or os.environ.get("ADVMIL_BAG_CACHE_GB") is not None
            scope = ("h", self.__dict__.setdefault("_cache_token", new_scope_token()), name_loader) if explicit else None
            view = loader_cache_view(self.device, train_loader, scope, None if gb is None else float(gb) * 1e9)
            if view is not None and view.scope == scope and scope not in self.__dict__.setdefault("_cache_scopes", set()):
                import weakref
                self._cache_scopes.add(scope)
                weakref.finalize(self, view.cache.drop_scope, scope)     # this handler's bags leave the device's cache with it
        seen = 0
        for bt in step_batches(train_loader, self.device, bp, view, drop_last=True, group_unstaged=True,
                               stageable=lambda x0: self.bcb != "graph",
                               pad_multiple=int(os.environ.get("ADVMIL_SLAB_PAD", self.cfg.get("slab_pad", 256))),
                               x_storage=self.x_storage):
            seen += len(bt.xs)
            if len(bt.xs) != bp:
                continue
            xs = bt.xs if bt.staged else [[dx.to(self.device, non_blocking=True) if torch.is_tensor(dx) else dx for dx in x] for x in bt.xs]
            if bt.staged and self.bcb == "cluster":
                xs = [[x[0]] + [dx.to(self.device, non_blocking=True) if torch.is_tensor(dx) else dx for dx in x[1:]] for x in xs]
            if all(not y.is_cuda for y in bt.ys):    # the step's labels: one pinned stack, one asynchronous copy
                y_pin = torch.empty(sum(y.shape[0] for y in bt.ys), bt.ys[0].shape[1], dtype=bt.ys[0].dtype, pin_memory=self.device.type == "cuda")
                torch.cat(bt.ys, dim=0, out=y_pin)
                y = y_pin.to(self.device, non_blocking=True)
            else:
                y = torch.cat([t.to(self.device) for t in bt.ys], dim=0)
            preds = self._update_network(seen, xs, y, bt.pad)
            ys_all.append(y); yh_all.append(preds)
        cltor = {"y": None, "y_hat": None}
        if ys_all:                                   # one D2H per key and epoch
            cltor = {"y": torch.cat(ys_all, dim=0).detach().cpu(), "y_hat": torch.cat(yh_all, dim=0).detach().cpu()}
        return cltor

    def _update_network(self, i_batch, xs, ys, pad=0):
        """One optimizer step over the collected bags (328-368): predictions [B, dim_out] of the step batch."""
        from .model_handler import MyHandler
        self.net.train()
        self.optimizer.zero_grad()
        X = MyHandler._slab_build_static(xs, True, pad)   # zero-copy when the bags sit back to back in the staging slab; operand planes attached
        lens = [x[0].shape[-2] for x in xs]
        seg = ops.Segments(lens + [pad] if pad else lens, self.device)      # (`pad` zero rows behind the bags: a dummy bag, dropped)
        exts = [x[1] for x in xs] if self.bcb in ("cluster", "graph") else None
        if exts is not None and pad:
            exts = exts + [torch.zeros(pad, dtype=exts[0].dtype, device=self.device)]
        preds = self.net.finish(self.net.features_multi(X, seg, exts)[:len(xs)])
        y = ys if torch.is_tensor(ys) else torch.cat(ys, dim=0)
        net_loss = self.supervised_loss(preds, y[:, 0:1], y[:, 1:2])
        with ops.deferred_sums():            # the backward's merge launches of parameter-gradient partials as one (ops.deferred_sums)
            net_loss.backward()
        total = net_loss.detach()
        if self.coef_l1 > 1e-8:
            total = total + self.coef_l1 * ops.abs_sum(self.optimizer.flat_param)[0]
        self.log({"train_batch/net/loss_supervision": net_loss.detach(), "train_batch/net/loss_total": total, "i_batch": i_batch})
        self.optimizer.step()
        return preds.detach()

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def test_model(model, backbone, loader, times_test_sample=1, checkpoint=None, batch_bags=None, x_storage=None):
        """443-487. The network has no noise, so in eval mode the `times_test_sample` repeated forwards of the reference are
        identical: a bag is run once and the sample axis is filled with that prediction. `batch_bags` (default 16,
        ADVMIL_EVAL_BATCH_BAGS) host bags are evaluated as one step slab, staged on the copy stream and kept in the device-resident
        bag cache between epochs (as MyHandler.test_model); device tensors, graphs and batch_bags=1 take the per-bag loop."""
        from ..ingest import loader_cache_view, step_batches
        from .model_handler import MyHandler
        dev = next(model.parameters()).device
        if checkpoint is not None:
            model.load_state_dict(torch.load(checkpoint, map_location=dev)["model"])
        model.eval()
        nb = int(batch_bags if batch_bags is not None else os.environ.get("ADVMIL_EVAL_BATCH_BAGS", "16"))
        if backbone == "graph" or not hasattr(model, "features_multi"):
            nb = 1
        idxs, ys, preds = [], [], []
        with torch.no_grad():
            for bt in step_batches(loader, dev, nb, loader_cache_view(dev, loader), pad_multiple=int(os.environ.get("ADVMIL_SLAB_PAD", "256")),
                                   x_storage=x_storage if x_storage is not None else getattr(model, "_advmil_x_storage", None)):
                if bt.staged:
                    xs, pad = bt.xs, bt.pad
                    X = MyHandler._slab_build_static(xs, True, pad)
                    lens = [x[0].shape[-2] for x in xs]
                    seg = ops.Segments(lens + [pad] if pad else lens, dev)
                    exts = [x[1].to(dev) if torch.is_tensor(x[1]) else x[1] for x in xs] if backbone == "cluster" else None
                    if exts is not None and pad:
                        exts = exts + [torch.zeros(pad, dtype=exts[0].dtype, device=dev)]
                    y_hat = model.finish(model.features_multi(X, seg, exts)[:len(xs)])
                else:
                    x_data, x_ext = [t.to(dev) if torch.is_tensor(t) else t for t in bt.xs[0]]
                    if backbone == "graph":
                        y_hat = model(x_ext, None)
                    elif backbone == "patch":
                        y_hat = model(x_data, None)
                    else:
                        y_hat = model(x_data, x_ext)
                idxs.extend(i.detach() for i in bt.idx); ys.extend(y.detach() for y in bt.ys); preds.append(y_hat)
        res = {"idx": None, "y": None, "y_hat": None}
        if preds:                                    # one D2H per key
            y_hat = torch.cat(preds, dim=0).detach().cpu()
            res = {"idx": torch.cat(idxs, dim=0).cpu(), "y": torch.cat(ys, dim=0).cpu(), "y_hat": y_hat}
            if times_test_sample > 1:
                dist = y_hat.unsqueeze(0).expand(times_test_sample, *y_hat.shape).contiguous()
                res["dist_y_hat"] = dist.transpose(0, 1).contiguous()
                res["avg_y_hat"] = torch.median(dist, dim=0)[0]
        return res

    # ------------------------------------------------------------------------------------------
    def _get_state_dict(self, epoch):
        return {"epoch": epoch, "model": self.net.state_dict(), "optimizer": self.optimizer.state_dict()}

    @staticmethod
    def _prefixed(path, prefix):
        d, f = osp.split(path)
        return osp.join(d, prefix + "_" + f)

    def _ckpt(self, ckpt_type, run_name):
        if ckpt_type not in ("best", "last"):
            raise KeyError("Expected best or last for `ckpt_type`, but got {}.".format(ckpt_type))
        return self._prefixed(self.last_net_ckpt_path if ckpt_type == "last" else self.best_net_ckpt_path, run_name)

    def save_model(self, epoch, ckpt_type="best", run_name="train"):
        torch.save(self._get_state_dict(epoch), self._ckpt(ckpt_type, run_name))

    def resume_model(self, ckpt_type="best", run_name="train"):
        ck = torch.load(self._ckpt(ckpt_type, run_name), map_location=self.device)
        self.net.load_state_dict(ck["model"])
        self.optimizer.load_state_dict(ck["optimizer"])

    def exec(self):
        raise NotImplementedError("orchestration (data loading, evaluator, early stopping, k-fold) stays with the reference: "
                                  "see INTEGRATION.md for the binding of _train_each_epoch / _update_network / test_model")
