"""The reference-side binding of INTEGRATION.md, dry-run in the build container (tests/golden/gen_golden_binding.py): the snippet
was EXECUTED over the imported reference with a recording stub in place of advmil_amd.model, and the reference's own
`_run_training` / `_eval_all` ran on top of it (model/model_handler.py:226-299, 500-569). The recorded trace -- every call the
reference's orchestration makes into the handler, with argument and return formats -- is the fixture
tests/golden/binding_trace.json. CPU: the real class accepts every recorded call. GPU: the real handler answers in exactly the
recorded formats (keys, dtypes, ranks, devices), so what `_eval_and_print` / `save_prediction` consumed in the dry run is what
they get from the HIP path.

init_weights (model/model_utils.py:12-17, applied as in model_handler.py:81): per-tensor checksums of the REFERENCE's generators
under a fixed torch seed (tests/golden/init_weights_v1.json) equal this repo's construction bit for bit."""
import inspect
import json
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _trace():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "binding_trace.json")))


def test_recorded_calls_bind_to_the_real_handler():
    from advmil_amd.model import MyHandler
    t = _trace()
    calls = [c["call"] for c in t["trace"]]
    assert calls[0] == "MyHandler.__init__" and "_train_each_epoch" in calls and "pop_logs" in calls and calls.count("test_model") == 3
    # the per-step methods stay bound for callers that drive them directly, with the reference's positional order
    for name, ref_sig in t["reference_signatures"].items():
        real = inspect.signature(getattr(MyHandler, name))
        ref_params = [p.strip().split("=")[0] for p in ref_sig.strip("()").split(",")]
        assert list(real.parameters)[:len(ref_params)] == ref_params, (name, list(real.parameters), ref_params)
    for c in t["trace"]:
        if c["call"] == "MyHandler.__init__":
            inspect.signature(MyHandler.__init__).bind(None, {k: None for k in c["cfg_keys"]})
            src = inspect.getsource(MyHandler.__init__) + inspect.getsource(MyHandler._train_each_epoch)
            import re
            needed = set(re.findall(r"cfg\[\"(\w+)\"\]", src)) - set(re.findall(r"cfg\.get\(\"(\w+)\"", src))   # (.get-guarded keys are optional)
            assert needed <= set(c["cfg_keys"]), needed - set(c["cfg_keys"])          # every key the ctor indexes is in cfg_nlst.yaml
        elif c["call"] in ("_train_each_epoch",):
            inspect.signature(MyHandler._train_each_epoch).bind(None, object(), c["args"]["name_loader"], mode=c["args"]["mode"])
        elif c["call"] == "test_model":
            a = c["args"]
            inspect.signature(MyHandler.test_model).bind(object(), object(), a["backbone"], object(), times_test_sample=a["times_test_sample"],
                                                         checkpoints=a["checkpoints"], test_zero_noise=a["test_zero_noise"])
        elif c["call"] == "pop_logs":
            inspect.signature(MyHandler.pop_logs).bind(None)
    assert set(t["checkpoint_files"]) == {"train_modelG-best.pth", "train_modelD-best.pth", "train_modelG-last.pth", "train_modelD-last.pth"}
    assert t["eval_all_returns"] == {"train": ["cindex", "loss"], "validation": ["cindex", "loss"]}
    # the wandb keys of the step log (model_handler.py:414-418, 489-494) came through the rebound _train_each_epoch
    for k in ("train_batch/netD/Loss_D", "train_batch/netD/D_real", "train_batch/netD/D_fake", "train_batch/netG/Loss_G_fake",
              "train_batch/netG/Loss_G_time", "train_batch/netG/Loss_G_total"):
        assert k in t["wandb_keys"], k


def test_init_weights_consume_the_references_random_stream():
    from types import SimpleNamespace
    from advmil_amd.model import Generator, load_backbone
    from advmil_amd.model.model_utils import init_weights
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "init_weights_v1.json")))
    for kind, want in fx["generators"].items():
        torch.manual_seed(fx["seed"])
        g = Generator(384, 1, load_backbone(kind, [1024, 384, 384]), SimpleNamespace(noise=[0, 1], hops=1, noise_dist="uniform"), False, 0.6,
                      "sigmoid")
        g.apply(init_weights)
        sd = g.state_dict()
        assert set(sd) == set(want), (kind, set(sd) ^ set(want))
        for k, w in want.items():
            v = sd[k]
            assert list(v.shape) == w["shape"], (kind, k)
            assert float(v.double().sum()) == w["sum"] and float(v.double().abs().sum()) == w["abs"], (kind, k)
            assert [float(x) for x in v.reshape(-1)[:4]] == w["head"], (kind, k)


def _fmt(v):
    return {"dtype": str(v.dtype).replace("torch.", ""), "rank": v.dim(), "device": v.device.type}


@pytest.mark.gpu
def test_real_handler_answers_in_the_recorded_formats(tmp_path):
    """The HIP handler on the dry run's own inputs (same item format, bp_every_batch 2, times_test_sample 5): collector keys,
    dtypes, ranks and devices equal what the reference's evaluator consumed; the logged keys equal the recorded ones; the
    reference's checkpoint layout round-trips through test_model(checkpoints=...)."""
    from advmil_amd.config import default_cfg
    from advmil_amd.model import MyHandler
    t = _trace()
    cfgd = t["cfg"]
    h = MyHandler(default_cfg(bcb_mode=cfgd["bcb_mode"], bp_every_batch=cfgd["bp_every_batch"], save_path=str(tmp_path)), device="cuda:0")
    h.patient_id.update({"label_visible": [f"p{i}" for i in range(4)], "train": [f"p{i}" for i in range(4)]})

    def bag(i, n=64):
        return (torch.tensor([[i]], dtype=torch.int), [torch.randn(1, n, 1024), torch.zeros(1, 1)], torch.tensor([[0.2 + 0.1 * i, float((i + 1) % 2)]]))

    rec = {c["call"]: c for c in t["trace"] if c["call"] != "test_model"}
    tm = [c for c in t["trace"] if c["call"] == "test_model"]
    train = [bag(i) for i in range(4)]
    cl = h._train_each_epoch(train, "train", "wlabel")
    want = rec["_train_each_epoch"]["returns"]
    assert set(cl) == set(want)
    for k, w in want.items():
        assert cl[k].shape[0] == w["tensor"][0] and _fmt(cl[k])["dtype"] == w["dtype"] and cl[k].device.type == "cpu", (k, cl[k].shape, w)
        assert list(cl[k].squeeze().shape) == [s for s in w["tensor"] if s != 1], k          # what the evaluator's .squeeze() sees
    logs = h.pop_logs()
    assert [sorted(d) for d in logs[:2]] == rec["pop_logs"]["returns"]
    h.save_model(1, ckpt_type="best", run_name="train")
    assert set(os.listdir(tmp_path)) >= {"train_modelG-best.pth", "train_modelD-best.pth"}
    for c in tm[:2]:
        a = c["args"]
        ck = None if a["checkpoints"] is None else [str(tmp_path / f) for f in a["checkpoints"]]
        res = MyHandler.test_model(h.netG, h.netD, a["backbone"], [bag(i) for i in range(a["loader"]["items"])],
                                   times_test_sample=a["times_test_sample"], checkpoints=ck, test_zero_noise=a["test_zero_noise"])
        assert set(res) == set(c["returns"])
        for k, w in c["returns"].items():
            assert list(res[k].shape) == w["tensor"] and _fmt(res[k])["dtype"] == w["dtype"] and res[k].device.type == "cpu", (k, res[k].shape, w)
