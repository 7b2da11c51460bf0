#!/usr/bin/env python3
"""Build container only (needs /root/reference; nothing here travels to the GPU box except the two JSON fixtures it writes).

1. binding_trace.json -- a DRY RUN of the reference-side binding of INTEGRATION.md. The rebinding snippet is cut out of
   INTEGRATION.md and executed over the IMPORTED reference (`model.model_handler.MyHandler`, shims of SURVEY.md Appendix A) with
   `advmil_amd.model` replaced by a recording stub (there is no GPU here, and the product has no CPU path): the stub handler
   holds this repo's real Generator / PrjDiscriminator modules (built on CPU, as the real ctor builds them before `.to(device)`),
   records every call the reference's orchestration makes into it and answers in the documented formats. The reference's own
   `_run_training` (one epoch, 4 training bags, 3 validation bags: `_train_each_epoch` -> `_eval_and_print` -> `test_model` ->
   `steplr.step` -> early stopping -> `save_model`) and `_eval_all` (checkpoint reload through `test_model(checkpoints=...)`,
   sampled predictions, evaluator, `save_prediction` patient-id lookup) then run for real on top of it
   (/root/reference/model/model_handler.py:226-299, 500-569). The recorded trace = what the reference hands to / expects from the
   HIP handler; tests/test_binding_trace_cpu.py checks it against the real class's signatures, tests/test_binding_trace_gpu.py
   checks that the real handler returns collectors of exactly the recorded format.
2. init_weights_v1.json -- per-tensor checksums of the reference's own generators right after `netG.apply(init_weights)` under
   a fixed torch seed (model_utils.py:12-17, model_handler.py:81): the product's init must consume the same random stream.

usage: python tests/golden/gen_golden_binding.py
"""
import inspect
import json
import os
import re
import sys
import tempfile
import types

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import gen_golden as GG  # noqa: E402  (shims; chdir to the reference)

TRACE = []


def summ(v):
    """Shape / dtype / device summary of an argument or a return value (what a binding must agree on)."""
    if torch.is_tensor(v):
        return {"tensor": list(v.shape), "dtype": str(v.dtype).replace("torch.", ""), "device": v.device.type}
    if isinstance(v, (list, tuple)):
        return [summ(x) for x in v[:3]] + (["..."] if len(v) > 3 else [])
    if isinstance(v, dict):
        return {k: summ(x) for k, x in v.items()}
    if isinstance(v, torch.nn.Module):
        return {"module": type(v).__name__}
    if v is None or isinstance(v, (bool, int, float, str)):
        return v
    return {"object": type(v).__name__}


def make_stub():
    from types import SimpleNamespace
    from advmil_amd.model import Discriminator, Generator, PrjDiscriminator, load_backbone
    from advmil_amd.model.model_utils import init_weights
    from advmil_amd.utils.func import agg_tensor, sparse_key, sparse_str

    class StubHandler:
        """Recording stand-in for advmil_amd.model.MyHandler (same ctor argument, same attributes the snippet touches)."""

        def __init__(self, cfg):
            TRACE.append({"call": "MyHandler.__init__", "cfg_keys": sorted(cfg)})
            backbone = load_backbone(cfg["bcb_mode"], sparse_str(cfg["bcb_dims"]))
            dim_in, dim_out = sparse_str(cfg["gen_dims"])
            an = SimpleNamespace(**sparse_key(cfg, prefixes="gen_noi"))
            an.noise = sparse_str(an.noise)
            self.netG = Generator(dim_in, dim_out, backbone, an, cfg["gen_norm"], cfg["gen_dropout"], cfg["gen_out_scale"])
            self.netG.apply(init_weights)
            dx = SimpleNamespace(**sparse_key(cfg, prefixes="disc_netx"))
            dy = SimpleNamespace(**sparse_key(cfg, prefixes="disc_nety"))
            dy.hid_dims = sparse_str(dy.hid_dims)
            cls = PrjDiscriminator if cfg["disc_type"] == "prj" else Discriminator
            self.netD = cls(dx, dy, prj_path=cfg["disc_prj_path"], inner_product=cfg["disc_prj_iprd"])
            # FlatAdam is a torch.optim.Optimizer with torch.optim.Adam's state_dict layout; the stand-in is the base class itself
            self.optimizerG = torch.optim.Adam(self.netG.parameters(), lr=cfg["opt_netG_lr"])
            self.optimizerD = torch.optim.Adam(self.netD.parameters(), lr=cfg["opt_netD_lr"])
            self.patient_id = {}
            self.cfg = cfg
            self._steps = 0

        def _train_each_epoch(self, train_loader, name_loader, mode="wlabel"):
            items = list(train_loader)
            TRACE.append({"call": "_train_each_epoch", "args": {"train_loader": {"items": len(items), "item": summ(items[0])},
                                                                  "name_loader": name_loader, "mode": mode}})
            bp = self.cfg["bp_every_batch"]
            n = len(items) // bp * bp
            self._steps = n // bp
            # the label-visibility lookup the step does per batch (reference model_handler.py:591-596) must work on the SHARED dict
            assert "label_visible" in self.patient_id and name_loader in self.patient_id
            cl = {"y": None, "y_hat": None, "f_fake": None}
            if n:
                cl = agg_tensor(cl, {"y": torch.cat([it[2] for it in items[:n]], dim=0).float(),
                                     "y_hat": torch.tensor([[0.25 + 0.1 * i] for i in range(n)], dtype=torch.float32),
                                     "f_fake": torch.zeros(n, dtype=torch.float32)})
            TRACE[-1]["returns"] = summ(cl)
            return cl

        def pop_logs(self):
            TRACE.append({"call": "pop_logs"})
            keys_d = ("train_batch/netD/Loss_D", "train_batch/netD/D_real", "train_batch/netD/D_fake")
            keys_g = ("train_batch/netG/Loss_G_fake", "train_batch/netG/Loss_G_time", "train_batch/netG/Loss_G_total",
                      "train_batch/netG/D_fake_avg")
            out = []
            for s in range(self._steps):
                out.append({**{k: 0.0 for k in keys_d}, "i_batch": (s + 1) * self.cfg["bp_every_batch"]})
                out.append({**{k: 0.0 for k in keys_g}, "i_batch": (s + 1) * self.cfg["bp_every_batch"]})
            TRACE[-1]["returns"] = [sorted(d) for d in out[:2]]
            return out

        def _update_disc(self, *a, **k):
            TRACE.append({"call": "_update_disc", "n_args": len(a), "kwargs": sorted(k)})
            raise AssertionError("the rebound _train_each_epoch never reaches the reference's per-step calls")

        _update_gen = _update_disc

        @staticmethod
        def test_model(modelG, modelD, backbone, loader, times_test_sample=1, checkpoints=None, test_zero_noise=False):
            items = list(loader)
            TRACE.append({"call": "test_model", "args": {"modelG": summ(modelG), "modelD": summ(modelD), "backbone": backbone,
                                                           "loader": {"items": len(items), "item": summ(items[0])},
                                                           "times_test_sample": times_test_sample,
                                                           "checkpoints": None if checkpoints is None else [os.path.basename(c) for c in checkpoints],
                                                           "test_zero_noise": test_zero_noise}})
            if checkpoints is not None:        # what the real test_model does first: the reference's save_model files must load
                modelG.load_state_dict(torch.load(checkpoints[0], map_location="cpu")["model"])
                modelD.load_state_dict(torch.load(checkpoints[1], map_location="cpu")["model"])
            res = {"idx": None, "y": None, "y_hat": None, "f_fake": None}
            for b, (idx, x, y) in enumerate(items):
                yh = torch.tensor([[0.3 + 0.1 * b]], dtype=torch.float32)
                res = agg_tensor(res, {"idx": idx.detach().cpu(), "y": y.detach().cpu(), "y_hat": yh, "f_fake": torch.zeros(1, 1)})
                if times_test_sample > 1:
                    ys = torch.stack([yh + 0.01 * k for k in range(times_test_sample)])
                    res = agg_tensor(res, {"dist_y_hat": ys.transpose(0, 1)})
                    res = agg_tensor(res, {"avg_y_hat": torch.median(ys, dim=0)[0]})
            TRACE[-1]["returns"] = summ(res)
            return res

    return types.SimpleNamespace(MyHandler=StubHandler)


def shim_class(stub):
    """The `class MyHandler(_RefHandler)` of INTEGRATION.md, executed as written; only its three import lines are redirected (the
    snippet lives in the reference's model/__init__.py, where the relative imports resolve; `hip` is the recording stub)."""
    text = open(os.path.join(REPO, "INTEGRATION.md")).read()
    src = next(b for b in re.findall(r"```python\n(.*?)```", text, re.S) if "class MyHandler(_RefHandler)" in b)
    src = src.replace('sys.path.insert(0, "/path/to/this/repo")', "pass")
    assert "import advmil_amd.model as hip" in src and "from .model_handler import MyHandler as _RefHandler" in src
    src = src.replace("import advmil_amd.model as hip", "hip = _STUB")
    src = src.replace("from .model_handler import MyHandler as _RefHandler", "from model.model_handler import MyHandler as _RefHandler")
    src = src.replace("from .baseline_handler import BaselineHandler", "from model.baseline_handler import BaselineHandler")
    ns = {"_STUB": stub}
    exec(compile(src, "INTEGRATION.md::rebinding-snippet", "exec"), ns)
    return ns["MyHandler"]


def dry_run():
    import model.model_handler as ref                       # the imported reference (sys.path via gen_golden's shims)
    stub = make_stub()
    Shim = shim_class(stub)
    cfg = yaml.load(open(os.path.join(REF, "config", "cfg_nlst.yaml")), Loader=yaml.FullLoader)
    tmp = tempfile.mkdtemp(prefix="advmil_binding_")
    cfg.update(bcb_mode="abmil", data_split_seed=0, save_path=os.path.join(tmp, "run"), wandb_dir=tmp, num_workers=0, bp_every_batch=2,
               epochs=1, times_test_sample=5, save_prediction=True, log_plot=False, es_patience=5, es_warmup=0, es_start_epoch=0)
    h = Shim(cfg)
    assert h.netG is h._hip.netG and h.netD is h._hip.netD and h.optimizerG is h._hip.optimizerG and h.optimizerD is h._hip.optimizerD
    assert h.steplr.optimizer is h._hip.optimizerG           # the reference's ReduceLROnPlateau drives the HIP optimizer's lr
    assert h._hip.patient_id is h.patient_id

    def bag(i, n=64):
        return (torch.tensor([[i]], dtype=torch.int), [torch.randn(1, n, 1024), torch.zeros(1, 1)], torch.tensor([[0.2 + 0.1 * i, float((i + 1) % 2)]]))

    train, val = [bag(i) for i in range(4)], [bag(i) for i in range(3)]
    h.patient_id.update({"label_visible": [f"p{i}" for i in range(4)], "train": [f"p{i}" for i in range(4)],
                         "validation": [f"p{i}" for i in range(3)]})
    n0 = len(GG.LOG)
    h._run_training(1, train, "train", val_loaders={"validation": val, "test": None}, val_name="validation",
                    measure_training_set=True, save_ckpt=True, early_stop=True, run_name="train")
    metrics = h._eval_all({"train": train, "validation": val, "test": None}, ckpt_type="best", run_name="train", if_print=True)
    files = sorted(os.listdir(cfg["save_path"]))
    wandb_keys = sorted({k for d in GG.LOG[n0:] for k in d})
    return {"reference": "liupei101/AdvMIL model/model_handler.py:226-299 (_run_training), 500-569 (_eval_all)",
            "cfg": {"bcb_mode": "abmil", "bp_every_batch": 2, "times_test_sample": 5, "epochs": 1},
            "trace": TRACE, "checkpoint_files": [f for f in files if f.endswith(".pth")], "prediction_files": [f for f in files if f.endswith(".csv")],
            "eval_all_returns": {k: [m[0] for m in v] for k, v in metrics.items()}, "wandb_keys": wandb_keys,
            "reference_signatures": {n: str(inspect.signature(getattr(ref.MyHandler, n))) for n in
                                     ("_train_each_epoch", "_update_disc", "_update_gen", "test_model", "save_model", "resume_model")}}


def init_checksums():
    """Per-tensor checksums of the reference's generators after `apply(init_weights)` under torch.manual_seed(1234)."""
    from types import SimpleNamespace
    from model.backbone import load_backbone
    from model.GANSurv import Generator
    from model.model_utils import init_weights
    out = {}
    for kind in ("abmil", "patch", "cluster"):
        torch.manual_seed(1234)
        g = Generator(384, 1, load_backbone(kind, [1024, 384, 384]), SimpleNamespace(noise=[0, 1], hops=1, noise_dist="uniform"), False, 0.6,
                      "sigmoid")
        g.apply(init_weights)
        out[kind] = {k: {"shape": list(v.shape), "sum": float(v.double().sum()), "abs": float(v.double().abs().sum()),
                         "head": [float(x) for x in v.reshape(-1)[:4]]} for k, v in g.state_dict().items()}
    return {"seed": 1234, "reference": "model/model_utils.py:12-17 applied as in model/model_handler.py:81", "generators": out}


def main():
    GG.install_shims()
    if not hasattr(np, "Inf"):
        np.Inf = np.inf                 # reference/numpy drift: utils/func.py:319 uses np.Inf (removed in NumPy 2.0); unrelated to the path
    res = dry_run()
    json.dump(res, open(os.path.join(HERE, "binding_trace.json"), "w"), indent=1)
    json.dump(init_checksums(), open(os.path.join(HERE, "init_weights_v1.json"), "w"), indent=1)
    print("calls:", [t["call"] for t in TRACE])
    print("ckpts:", res["checkpoint_files"], "preds:", res["prediction_files"], "metrics:", res["eval_all_returns"])


if __name__ == "__main__":
    main()
