"""MyHandler: the training-step caller of the reference (model/model_handler.py) -- ctor (37-137),
_train_each_epoch (301-347), _update_disc (349-424), _update_gen (426-498), test_model (598-643),
save_model / resume_model (645-678) -- with the same names, arguments, return values, logged keys and
checkpoint layout, driving the HIP path.

What is deliberately different from the reference's schedule (results identical):
  * the D embedding (FC+LN+ReLU+mean16 of the bag) is computed once per bag in the D phase and shared by the
    real and the fake pair (it has no dropout and does not depend on t);
  * the D-phase generator forward runs under no_grad (the reference builds then detaches its graph);
  * in the G phase nothing of D that touches the bag is differentiated (dD/dx is not a function of G's weights);
  * each bag's loss term is back-propagated as soon as it exists, scaled by the step-batch denominators the
    reference's means use, instead of keeping bp_every_batch autograd graphs alive;
  * no per-bag host syncs: event flags come from the host copy of the labels, logs stay on the device until the
    epoch ends; the bag is never copied by a boolean mask; no empty_cache() per step;
  * the L1 term's gradient is applied inside the fused Adam kernel; its value is still added to Loss_G_total.
Dataset / evaluator / wandb orchestration (exec, _run_training, _eval_all, exec_semi_sl) is out of scope
(SURVEY.md §2 #6, #9-11): see INTEGRATION.md for how the reference's own handler binds to this class.
"""
import os
import os.path as osp
from functools import partial
from types import SimpleNamespace

import torch

from .. import ops
from ..loss.utils import fake_generator_loss, real_fake_loss, real_fake_terms, recon_loss, recon_terms
from ..optim import FlatAdam, create_optimizer
from ..ingest import BagCache, SlabStager, bag_fingerprint, x_store_dtype
from ..parallel import BagParallel
from ..utils.func import agg_tensor, seed_everything, sparse_key, sparse_str
from .backbone import load_backbone
from .GANSurv import Discriminator, Generator, PrjDiscriminator
from .model_utils import init_weights


def _check_configs(cfg):
    """The constraints of model_handler.py:780-812 that concern the step."""
    assert cfg["task"] in ("cont_gansurv",), "HIP path covers task=cont_gansurv (the only task in the shipped config)"
    assert cfg["batch_size"] == 1, "batch_size must be 1 (one WSI per forward)"
    assert cfg["loss_netD"] in ("bce", "hinge", "wasserstein")
    assert cfg["disc_type"] in ("prj", "cat")
    assert cfg["gen_out_scale"] in ("sigmoid", "exp", "none", None)


class LazyLog:
    """A logged record whose numbers are derived on the host when the log is read: `tensors` are device tensors the step wrote
    (a HIP-graph replay rewrites them in place), `fn(*lists)` maps their values to the record."""

    def __init__(self, tensors, fn):
        self.tensors, self.fn = tuple(tensors), fn

    def resolve(self):
        import numpy as np
        vals = [np.asarray(t.detach().float().cpu().tolist(), dtype=np.float32) for t in self.tensors]     # fp32 arithmetic, as on the device
        out = self.fn(*vals)
        return {k: (float(v) if not isinstance(v, int) else v) for k, v in out.items()}


class StaticStepPlan:
    """A step plan (`MyHandler._plan`) whose device arrays live in ONE static allocation, so that the HIP graph captured over it can be
    replayed for ANY later step batch of the same KEY -- bags per step, rows of the (padded) slab, number of real pairs / visible labels,
    slab buffer -- after `rebind` has rewritten the arrays for that batch: bag lengths (segment offsets, per-row bag ids of the slab, of
    its 16-row regions and of the stacked fake | real regions), labels, real / visible masks, the stacked layout's dropout row map. One
    pinned buffer, one asynchronous copy per step. The reference's loop has no such state: it re-issues ~480 launches per bag
    (model_handler.py:311-345); here a ragged resident epoch replays one graph per step like the fixed-shape bench does.

    What may differ between the capture batch and a replayed one: only what lives in the arrays. What may NOT: everything in `key`.
    Grids of the segmented kernels are sized for the longest bag the key admits (`bag_cap`): workgroups past a bag's end exit at once."""

    @staticmethod
    def bag_cap(lens, pad, total):
        """Longest bag the launch grids of a key are sized for: the next power of two above the batch's longest bag (part of the key: a
        batch with a longer bag takes another graph). Sizing them for the whole slab instead cost 0.16 ms per 16-bag step in empty
        workgroups of the pooling backward (profiles/r06_step_profile_loop.txt)."""
        m = max(list(lens) + [int(pad), 1])
        return int(min(total, max(1024, 1 << (m - 1).bit_length())))

    def __init__(self, handler, plan, lens, pad, cap):
        import numpy as np
        self.n, self.dev, self.cap = len(lens), handler.device, int(cap)
        n, T = self.n, sum(lens) + int(pad)
        self.T, self.has_pad = T, bool(pad)
        ns = n + (1 if pad else 0)
        assert T % 16 == 0 and all(v % 16 == 0 for v in lens)
        self.has_t2 = plan.t2 is not None
        self.has_map = plan.rng_rows is not None and "region2" in plan.rng_rows
        # ---- layout (bytes, every array 16-byte aligned)
        self.lay, off = {}, 0
        def put(name, count, dt):
            nonlocal off
            self.lay[name] = (off, count, dt)
            off += (count * np.dtype(dt).itemsize + 15) // 16 * 16
        put("y", 2 * n, np.float32); put("y_t", n, np.float32); put("y_e", n, np.float32)
        if self.has_t2:
            put("t2", 2 * n, np.float32)
        put("masks", 2 * n, np.float32)
        put("seg_ptr", ns + 1, np.int64); put("seg16_ptr", ns + 1, np.int64); put("x2_ptr", 2 * ns + 1, np.int64)
        if self.has_map:
            put("region2", 2 * (T // 16), np.int64)
        put("seg_row", T, np.int32); put("seg16_row", T // 16, np.int32); put("x2_row", 2 * (T // 16), np.int32)
        self.nbytes = off
        self.buf = torch.empty(self.nbytes, dtype=torch.uint8, device=self.dev)
        tdt = {np.float32: torch.float32, np.int64: torch.int64, np.int32: torch.int32}
        def view(name):
            o, c, dt = self.lay[name]
            return self.buf[o:o + c * np.dtype(dt).itemsize].view(tdt[dt])
        # ---- the plan's arrays become views of the static buffer (addresses baked by the capture)
        plan.y = view("y").view(n, 2)
        plan.y_t, plan.y_e = view("y_t").view(n, 1), view("y_e").view(n, 1)
        if self.has_t2:
            plan.t2 = view("t2").view(2 * n, 1)
        m = view("masks")
        plan.real_mask = m[:n]
        if plan.vis_mask is not None:
            plan.vis_mask = m[n:]
        seg, seg16 = plan.seg, plan.seg16
        x2 = seg16.twice()
        for sg, pn, rn, cap_ in ((seg, "seg_ptr", "seg_row", self.cap), (seg16, "seg16_ptr", "seg16_row", self.cap // 16),
                                 (x2, "x2_ptr", "x2_row", self.cap // 16)):
            sg.ptr, sg.rowseg = view(pn), view(rn)
            sg.max_len = cap_                # grids for the longest bag the key admits (bag_cap)
            sg.lens = sg.offsets = None      # (host-side copies would go stale: nothing on the ABMIL path reads them after construction)
            sg._div = {k: v for k, v in sg._div.items() if k != "long"}
        if self.has_map:
            plan.rng_rows["region2"] = view("region2")
        # (plan.vis / plan.is_real stay the capture batch's lists: the step only asks any(vis), which the key's counts fix)
        self.plan = plan

    def rebind(self, lens, pad, ys_host, is_real, vis, stream=None):
        """Rewrite the arrays for a batch of the same key: ONE pinned buffer, ONE asynchronous copy (on `stream`, default the current one)."""
        import numpy as np
        n, T = self.n, self.T
        assert len(lens) == n and sum(lens) + int(pad) == T and bool(pad) == self.has_pad and max(list(lens) + [int(pad)]) <= self.cap
        host = torch.empty(self.nbytes, dtype=torch.uint8, pin_memory=True)      # (the caching host allocator keeps it until the copy ran)
        hv = host.numpy()
        def arr(name):
            o, c, dt = self.lay[name]
            return hv[o:o + c * np.dtype(dt).itemsize].view(dt)
        y = torch.cat([h.reshape(1, 2).float() for h in ys_host], dim=0).numpy()
        arr("y")[:] = y.reshape(-1)
        arr("y_t")[:] = y[:, 0]
        arr("y_e")[:] = y[:, 1]
        if self.has_t2:
            t2 = arr("t2")
            t2[:n] = 0.0                     # (rows [0, n): written by the generator's head inside the step)
            t2[n:] = y[:, 0]
        mk = arr("masks")
        mk[:n] = [1.0 if q else 0.0 for q in is_real]
        mk[n:] = [1.0 if v else 0.0 for v in vis]
        ls = list(lens) + ([int(pad)] if pad else [])
        l16 = [v // 16 for v in ls]
        for pn, rn, ll in (("seg_ptr", "seg_row", ls), ("seg16_ptr", "seg16_row", l16), ("x2_ptr", "x2_row", l16 + l16)):
            a = np.asarray(ll, dtype=np.int64)
            p = arr(pn)
            p[0] = 0
            np.cumsum(a, out=p[1:])
            arr(rn)[:] = np.repeat(np.arange(len(ll), dtype=np.int32), a)
        if self.has_map:                     # MyHandler._rng_row_maps, W = 1 with a pad
            L, p16 = sum(lens) // 16, (int(pad) + 15) // 16
            ar, ap = np.arange(L, dtype=np.int64), np.arange(p16, dtype=np.int64)
            arr("region2")[:] = np.concatenate([ar, 2 * L + ap, L + ar, 2 * L + p16 + 1 + ap])
        if stream is None:
            self.buf.copy_(host, non_blocking=True)
        else:
            with torch.cuda.stream(stream):
                self.buf.copy_(host, non_blocking=True)


class MyHandler(object):
    def __init__(self, cfg, device=None, parallel=None):
        _check_configs(cfg)
        if device is None:
            ndev = torch.cuda.device_count()
            if ndev == 0:
                raise RuntimeError("advmil_amd.MyHandler needs an MI355X: no ROCm device visible (no CPU fallback)")
            local = int(os.environ.get("LOCAL_RANK", cfg.get("cuda_id", 0)))
            device = torch.device("cuda", local % ndev)
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        seed_everything(cfg["seed"])
        self.rng = ops.default_rng(self.device)
        self.dp = parallel or BagParallel()
        # (The generator's training forward as a parallel stream / graph branch beside the discriminator's backward was built in round 3 and
        # measured slower at every step size -- the slab kernels of both branches each want every CU's LDS --; retired in round 6,
        # docs/DESIGN_HISTORY.md.)
        # rows of a staged step slab are padded with zero rows to a multiple of this (0 = off): see ingest.SlabStager.pad_rows
        self.slab_pad = int(os.environ.get("ADVMIL_SLAB_PAD", cfg.get("slab_pad", 256)))
        # shape-keyed step graphs of the epoch loop (StaticStepPlan): cfg['step_graphs'] / ADVMIL_STEP_GRAPHS (1), the row granularity of
        # their keys, and how many keys are kept
        self.step_graphs = str(os.environ.get("ADVMIL_STEP_GRAPHS", cfg.get("step_graphs", 1))) != "0"
        self.step_graph_rows = int(os.environ.get("ADVMIL_STEP_GRAPH_ROWS", cfg.get("step_graph_rows", 2048)))
        self.step_graphs_max = int(os.environ.get("ADVMIL_STEP_GRAPHS_MAX", cfg.get("step_graphs_max", 24)))
        self._step_graph_cache, self._step_graph_seen, self._step_graph_pool = {}, {}, None
        self.step_graph_stats = {"replayed": 0, "captured": 0, "eager": 0}
        # 'fp32' (default) | 'bf16': how bags are held in HBM (staging slab, device-resident cache). 'bf16' = ONE bf16 plane per bag
        # (half the bytes; the contractions over the slab form each product with two MFMAs instead of three); see DESIGN.md
        self.x_storage = os.environ.get("ADVMIL_X_STORAGE", cfg.get("x_storage", "fp32"))
        self.cfg = cfg
        self.bcb = cfg["bcb_mode"]
        self.task = cfg["task"]

        save_path = cfg.get("save_path")
        if save_path:
            os.makedirs(save_path, exist_ok=True)
            self.last_netD_ckpt_path = osp.join(save_path, "modelD-last.pth")
            self.best_netD_ckpt_path = osp.join(save_path, "modelD-best.pth")
            self.last_netG_ckpt_path = osp.join(save_path, "modelG-last.pth")
            self.best_netG_ckpt_path = osp.join(save_path, "modelG-best.pth")

        # ---- networks (model_handler.py:70-91)
        backbone = load_backbone(self.bcb, sparse_str(cfg["bcb_dims"]))
        dim_in, dim_out = sparse_str(cfg["gen_dims"])
        args_noise = SimpleNamespace(**sparse_key(cfg, prefixes="gen_noi"))
        args_noise.noise = sparse_str(args_noise.noise)
        self.netG = Generator(dim_in, dim_out, backbone, args_noise, cfg["gen_norm"], cfg["gen_dropout"], cfg["gen_out_scale"])
        self.netG.apply(init_weights)                      # G: xavier; D keeps torch defaults (model_handler.py:81)
        disc_x = SimpleNamespace(**sparse_key(cfg, prefixes="disc_netx"))
        disc_y = SimpleNamespace(**sparse_key(cfg, prefixes="disc_nety"))
        disc_y.hid_dims = sparse_str(disc_y.hid_dims)
        cls = PrjDiscriminator if cfg["disc_type"] == "prj" else Discriminator
        self.netD = cls(disc_x, disc_y, prj_path=cfg["disc_prj_path"], inner_product=cfg["disc_prj_iprd"])
        self.netG = self.netG.to(self.device)
        self.netD = self.netD.to(self.device)
        for m in list(self.netG.modules()) + list(self.netD.modules()):
            m.rng = self.rng
        # the STATIC test_model (reference signature: no handler argument) reads the bags' storage mode off the generator it is given, so
        # a cfg-only x_storage = 'bf16' is validated / tested on the same rounded bags it trains on, out of the same device cache entries
        self.netG._advmil_x_storage = self.x_storage
        # arithmetic of the fp32 contraction engine (process-wide switch in the library): "exact" = fp32 MFMA,
        # "bf16x3" = split-bf16 on the bf16 matrix pipe with fp32 accumulate (near-fp32: ~2^-17 per product). None = leave as is.
        if cfg.get("gemm_mode") is not None:
            ops.set_gemm_mode(cfg["gemm_mode"])
        if self.dp.world > 1:                              # replicas start identical
            for net in (self.netG, self.netD):
                for p in net.parameters():
                    self.dp.broadcast_(p.data)

        # ---- losses / optimizers (model_handler.py:94-109)
        self.which_loss = cfg["loss_netD"]
        self.real_fake_loss = partial(real_fake_loss, which=cfg["loss_netD"])
        self.supervised_loss = partial(recon_loss, **sparse_key(cfg, prefixes="loss_recon"))
        self.supervised_terms = partial(recon_terms, **sparse_key(cfg, prefixes="loss_recon"))
        self._recon = dict(alpha=0.0, gamma=1.0, norm="l1")
        self._recon.update(sparse_key(cfg, prefixes="loss_recon"))
        self.coef_ganloss = cfg["loss_gan_coef"]
        self.coef_l1 = 0.0 if cfg["loss_regl1_coef"] is None else float(cfg["loss_regl1_coef"])
        opt_cfg = SimpleNamespace(opt=cfg["opt_netG"], weight_decay=cfg["opt_netG_weight_decay"], lr=cfg["opt_netG_lr"],
                                  opt_eps=None, opt_betas=None, momentum=None)
        self.optimizerG = create_optimizer(opt_cfg, self.netG)
        self.optimizerG.l1_coef = self.coef_l1 if self.coef_l1 > 1e-8 else 0.0
        self.optimizerD = FlatAdam(self.netD, lr=cfg["opt_netD_lr"], betas=(0.9, 0.999), weight_decay=0.0)
        self.steplr = torch.optim.lr_scheduler.ReduceLROnPlateau(self.optimizerG, mode="min", factor=0.5, patience=10)

        self._one_t = torch.ones((), dtype=torch.float32, device=self.device)     # the root gradient of both losses
        self.patient_id = dict()
        self.history = []          # list of dicts of DEVICE scalars, flushed by pop_logs()
        self.epoch = 0
        self.noise_hook = None     # tests: callable(phase 'd'|'g', bag idx) -> [noise tensors] injected into G

    # ------------------------------------------------------------------------------------------
    def log(self, d):
        self.history.append(d)

    @staticmethod
    def resolve_log(d):
        """One history entry -> dict of python numbers. An entry is a dict of device scalars, or a LazyLog: the step's statistics
        tensor(s) plus the host arithmetic that turns them into the logged quantities (the divisions by the global counts, the
        L1 term, the sign flip), evaluated here instead of as four or five one-element launches inside every optimizer step."""
        if isinstance(d, LazyLog):
            return d.resolve()
        return {k: (float(v) if torch.is_tensor(v) else v) for k, v in d.items()}

    def pop_logs(self):
        """Device scalars -> python floats (one sync for the whole backlog)."""
        out = [self.resolve_log(d) for d in self.history]
        self.history = []
        return out

    def _get_patient_id(self, k, idxs):
        pids = self.patient_id[k]
        return [pids[i] for i in idxs.reshape(-1).tolist()]

    @staticmethod
    def _set_mode(net, training):
        """net.train(training) without the recursive Module.train walk (4 walks per optimizer step x ~30 modules through
        Module.__setattr__: 0.25-0.5 ms of host time per step): `training` is a plain attribute of every module."""
        mods = net.__dict__.get("_advmil_modules")
        if mods is None or net.__dict__.get("_advmil_modules_n") != sum(1 for _ in net.children()):
            mods = list(net.modules())
            net.__dict__["_advmil_modules"] = mods
            net.__dict__["_advmil_modules_n"] = sum(1 for _ in net.children())
        for m in mods:
            m.__dict__["training"] = training

    def _get_label_visiable_mask(self, k, idxs):
        if "label_visible" not in self.patient_id:
            return None
        if isinstance(idxs, list):
            idxs = torch.cat(idxs, dim=0)
        visible = set(self.patient_id["label_visible"])
        return [p in visible for p in self._get_patient_id(k, idxs)]

    def _gen_forward(self, data_x, data_x_ext, **kw):
        if self.bcb == "graph":
            return self.netG(data_x_ext, None, **kw)
        if self.bcb == "patch":
            return self.netG(data_x, None, **kw)       # coords skipped (model_handler.py:390)
        return self.netG(data_x, data_x_ext, **kw)

    @staticmethod
    def _vis(mode, n, label_visible_mask):
        if label_visible_mask is None:
            return [mode == "wlabel"] * n
        return [mode == "wlabel" or (mode == "wolabel" and bool(m)) for m in label_visible_mask]

    # ------------------------------------------------------------------------------------------
    def _train_each_epoch(self, train_loader, name_loader, mode="wlabel"):
        """Same contract as the reference (model_handler.py:301-347). Bags arriving as CPU tensors are staged through the
        double-buffered pinned slab (advmil_amd/ingest.py): async H2D on a copy stream, the step batch contiguous in HBM.
        Under bag-parallel every rank walks its own shard of the loader (bag i of the global step batch on rank i mod W) and the
        returned collector is all-gathered back into global bag order (model_handler.py:333-339 semantics)."""
        # cfg['bp_every_batch'] is the GLOBAL step batch (cfg_nlst.yaml:71) at any world size: this rank steps every bp / W bags
        bp_every_batch = self.dp.local_step_bags(self.cfg["bp_every_batch"])
        num_update_gen = self.cfg["gen_updates"]
        if self.dp.world > 1 and hasattr(train_loader, "__len__"):
            self.dp.check_equal(len(train_loader) // bp_every_batch, "the number of optimizer steps of this epoch", self.device)
        ys_all, yhat_all, ffake_all = [], [], []
        i_col, x_col, y_col, yh_col = [], [], [], []
        stager = None
        staged = False
        cache = self._bag_cache_for(name_loader, train_loader)
        fresh = []                               # (cache key, index into x_col) of the bags of this step that came over PCIe
        staged_pos = []                          # indices into x_col of the bags of this step that sit in the staging slab
        i_batch = 0
        # Every optimizer step numbers its dropout / noise call sites from the same base and advances the device seed by one behind it --
        # what a captured step does by itself (graphed.GraphedStep) -- so that an eager step and a replayed one draw the same masks. The
        # base is fixed with the handler's first epoch (a replay leaves the host-side counter where it was: it must not drift with it)
        if getattr(self, "_site_base", None) is None:
            self._site_base = self.rng.counter
        site_base = self._site_base
        for data_idx, data_x, data_y in train_loader:
            i_batch += 1
            yh_col.append(data_y if not data_y.is_cuda else None)
            x0 = data_x[0]
            if torch.is_tensor(x0) and not x0.is_cuda and self.bcb != "graph":
                key = int(data_idx.reshape(-1)[0]) if cache is not None else None
                fp = bag_fingerprint(x0) if cache is not None else None
                hit = cache.get(key, fp) if cache is not None else None
                if stager is None:
                    stager = self.__dict__.setdefault("_stager", None)
                    store = x_store_dtype(self.x_storage, x0.dtype)
                    if stager is None or stager.dtype != x0.dtype or stager.store != store or stager.channels != x0.shape[-1]:
                        stager = SlabStager(self.device, x0.shape[-1], x0.dtype, store)
                    self._stager = stager
                if hit is not None and hit.dtype != stager.store:      # kept under another x_storage: not this slab's bag
                    hit = None
                if not staged:
                    stager.expect(bp_every_batch, x0.shape[1])
                    stager.begin()
                    staged = True
                staged_pos.append(len(x_col))
                if hit is not None:              # resident since an earlier epoch: no H2D; its rows (and operand planes) are copied into
                    #                              the step slab device to device on the copy stream, under the previous step's compute
                    # (the entry keeps its event: any copy stream that reads it is ordered behind the kernels that made it)
                    v = stager.add_device(hit, hit.__dict__.get("_advmil_bag_planes"), hit.__dict__.get("_advmil_ready"))
                else:
                    if cache is not None:
                        fresh.append((key, len(x_col), fp))
                    v = stager.add(x0)
                # (the second loader field is the cluster ids / graph of those two backbones; ABMIL and ESAT never read it, and a pageable
                # host tensor's `.to()` is a host-synchronous copy that would throttle the launch queue to the device's pace)
                # DeepAttMISL's cluster ids stay on the host until the step: `_gen_features` sends the step's ids as ONE pinned buffer
                ext_used = self.bcb == "graph"
                data_x = [v] + [dx.to(self.device, non_blocking=True) if (ext_used and torch.is_tensor(dx)) else dx for dx in data_x[1:]]
            else:
                data_x = [dx.to(self.device, non_blocking=True) if torch.is_tensor(dx) else dx for dx in data_x]
            i_col.append(data_idx); x_col.append(data_x); y_col.append(data_y)
            if i_batch % bp_every_batch == 0:
                ys_host = None if any(h is None for h in yh_col) else yh_col
                if ys_host is not None and len({tuple(h.shape) for h in ys_host}) == 1 and ys_host[0].dim() == 2:
                    # the step's labels in ONE pinned stack and one asynchronous copy (was: one tiny H2D per bag + two concatenations)
                    y_pin = torch.empty(sum(h.shape[0] for h in ys_host), ys_host[0].shape[1], dtype=ys_host[0].dtype, pin_memory=True)
                    torch.cat(ys_host, dim=0, out=y_pin)
                    y_step = y_pin.to(self.device, non_blocking=True)
                    y_col = list(y_step.split([h.shape[0] for h in ys_host], dim=0))
                else:
                    y_col = [y.to(self.device, non_blocking=True) for y in y_col]
                    y_step = torch.cat(y_col, dim=0)
                pad = 0
                bpl = None
                if staged:                               # growth may have re-based the views: take the final ones
                    if len(staged_pos) == len(x_col):
                        # whole 256-row tiles for the slab kernels' fast forms; with step graphs on, whole `step_graph_rows` (2048): the
                        # padded row count is part of a graph's key, so coarser steps mean fewer captures (at most 1.6 % more rows at 16 x 8192)
                        sg_on = self.step_graphs and self.dp.world == 1 and self.bcb == "abmil" and self.slab_pad > 0
                        gran = self.slab_pad
                        if sg_on:            # <= 1.6 % more rows: 256 below 32768 rows, 2048 (step_graph_rows) from 131072 rows on
                            while gran * 2 <= self.step_graph_rows and gran * 2 * 64 <= stager.rows:
                                gran *= 2
                        pad = stager.pad_rows(gran)
                    # (a batch that mixes staged and device bags is concatenated, i.e. read as fp32 rows)
                    for j, v in zip(staged_pos, stager.ready(need_rows=len(staged_pos) != len(x_col))):
                        x_col[j][0] = v
                    bpl = stager.batch_planes()          # every bag came from the cache with its planes: the slab's planes are ready
                    if bpl is not None and staged_pos:
                        x_col[staged_pos[0]][0]._advmil_stager_planes = bpl
                        if stager.stale:                 # cached bags staged as operand planes only: the slab's fp32 rows are not valid
                            x_col[staged_pos[0]][0]._advmil_fp32_stale = True
                mask = self._get_label_visiable_mask(name_loader, i_col)
                nz_d = nz_g = None
                if self.noise_hook is not None:
                    nz_d = [self.noise_hook("d", int(ix.reshape(-1)[0])) for ix in i_col]
                    nz_g = [self.noise_hook("g", int(ix.reshape(-1)[0])) for ix in i_col]
                # ---- shape-keyed step graphs (round 6): a resident ragged step batch whose KEY has been seen before is REPLAYED -- the
                # static plan's arrays are rewritten for this batch (one small copy), then one graph launch instead of ~53 eager ones
                replayed = False
                gkey = cap_ = None
                self.rng.counter = site_base
                if staged and len(staged_pos) == len(x_col) and self.step_graphs and self.dp.world == 1 and self.bcb == "abmil":
                    # launch grids of the segmented kernels for the longest bag a step-graph key admits: for EVERY staged batch of this
                    # loop, graph-eligible or not, so that eager and replayed steps (and fp32 / bf16 bag storage) share one geometry
                    lens_ = [self._rows(x[0]) for x in x_col]
                    cap_ = StaticStepPlan.bag_cap(lens_, pad, sum(lens_) + pad)
                if (staged and len(staged_pos) == len(x_col) and self.step_graphs and self.dp.world == 1 and self.bcb == "abmil"
                        and self.noise_hook is None and num_update_gen == 1 and ys_host is not None and bpl is not None
                        and not torch.cuda.is_current_stream_capturing()):
                    n_ = len(x_col)
                    vis_ = self._vis(mode, n_, mask)
                    real_ = [bool(float(yh[0, 1]) == 1.0) and vis_[i] for i, yh in enumerate(ys_host)]
                    x0_ = x_col[0][0]
                    gkey = (self.bcb, n_, sum(lens_) + pad, cap_, int(x0_.shape[-1]), x0_.dtype, x0_.data_ptr(), bool(pad), sum(real_), n_, sum(vis_),
                            all(vis_), mode, bool(stager.stale), ops.get_gemm_mode(), self.optimizerG.param_groups[0]["lr"],
                            self.optimizerD.param_groups[0]["lr"])
                    ent = self._step_graph_cache.get(gkey)
                    if ent is not None:
                        sp, g = ent
                        sp.rebind(lens_, pad, ys_host, real_, vis_)
                        g.replay()
                        self.step_graph_stats["replayed"] += 1
                        preds_t, fakes_t = torch.cat(g.preds, dim=0).detach(), torch.cat(g.fakes, dim=0).detach()
                        # this step's logged statistics: the graph rewrites its own tensors with every replay -> one snapshot launch
                        tens = [t for lg in g.logs for t in lg.tensors]
                        flat = torch.cat([t.detach().reshape(-1).float() for t in tens])
                        o = 0
                        for lg in g.logs:
                            snap = []
                            for t in lg.tensors:
                                snap.append(flat[o:o + t.numel()].view(t.shape)); o += t.numel()
                            self.log(LazyLog(snap, (lambda *v, f=lg.fn, ib=i_batch: {**f(*v), "i_batch": ib})))
                        replayed = True
                    else:
                        seen = self._step_graph_seen.get(gkey, 0) + 1
                        self._step_graph_seen[gkey] = seen
                        if seen >= 2 and len(self._step_graph_cache) < self.step_graphs_max:
                            # second sight of this key: THIS batch's step is the capture's warm-up step (a real step), then the capture
                            from ..graphed import GraphedStep
                            plan = self._plan(x_col, y_col, mode, mask, ys_host, y_step, pad)
                            sp = StaticStepPlan(self, plan, lens_, pad, cap_)
                            sp.rebind(lens_, pad, ys_host, real_, vis_)
                            if self._step_graph_pool is None:
                                self._step_graph_pool = torch.cuda.graph_pool_handle()
                            g = GraphedStep(self, x_col, y_col, ys_host, mode, mask, warmup=1, plan=plan, keep_warmup=True,
                                            pool=self._step_graph_pool, site_base=site_base)
                            self._step_graph_cache[gkey] = (sp, g)
                            self.step_graph_stats["captured"] += 1
                            for q in (-2, -1):   # (the warm-up step's two records were logged under the capture's batch number 0)
                                lg_ = self.history[q]
                                if isinstance(lg_, LazyLog):
                                    self.history[q] = LazyLog(lg_.tensors, (lambda *v, f=lg_.fn, ib=i_batch: {**f(*v), "i_batch": ib}))
                            preds_t, fakes_t = g.warm_out
                            replayed = True
                if not replayed:
                    plan = self._plan(x_col, y_col, mode, mask, ys_host, y_step, pad)   # ONE plan per step batch, shared by the D and G updates
                    if cap_ is not None:     # (the same launch grids as a step graph's capture: StaticStepPlan)
                        plan.seg.max_len, plan.seg16.max_len, plan.seg16.twice().max_len = cap_, cap_ // 16, cap_ // 16
                    # bag-parallel: D's gradient exchange is started asynchronously and completed inside the first generator update, after
                    # the generator's backbone forward (which does not depend on D) has been enqueued -> the two overlap
                    overlap = self.dp.world > 1 and num_update_gen > 0
                    preds, fakes = self._update_disc(i_batch, x_col, y_col, mode, mask, ys_host=ys_host, noise=nz_d, plan=plan, defer_apply=overlap)
                    for _ in range(num_update_gen):
                        self._update_gen(i_batch, x_col, y_col, mode, mask, ys_host=ys_host, noise=nz_g, plan=plan)
                    preds_t, fakes_t = torch.cat(preds, dim=0).detach(), torch.cat(fakes, dim=0)
                    self.rng.advance(1)
                    self.step_graph_stats["eager"] += 1
                for key, j, fp in fresh:                 # first sight of these bags: keep them (and their operand planes) in HBM
                    cache.put(key, x_col[j][0], fp)
                fresh, staged_pos = [], []
                if staged:
                    stager.release()
                    staged = False
                ys_all.append(y_step); yhat_all.append(preds_t)
                ffake_all.append(fakes_t)
                i_col, x_col, y_col, yh_col = [], [], [], []
        cltor = {"y": None, "y_hat": None, "f_fake": None}
        if ys_all:                                      # one D2H per epoch instead of one per step
            gather = self.dp.allgather_cat                # identity at world == 1; global bag order (j * W + r) otherwise
            cltor = agg_tensor(cltor, {"y": gather(torch.cat(ys_all)).cpu(), "y_hat": gather(torch.cat(yhat_all)).cpu(),
                                       "f_fake": gather(torch.cat(ffake_all)).cpu()})
        return cltor

    def _bag_cache_for(self, name_loader, loader=None):
        """This loader's view of the device-resident bag cache (advmil_amd/ingest.py::BagCache, ONE per device, shared with the
        evaluation passes), or None. Budget: cfg['bag_cache_gb'] / ADVMIL_BAG_CACHE_GB (0 = off); default 30 % of the device's
        memory. Scope of the keys: the loader's dataset object when it has one (DataLoader), else (this handler, name_loader)."""
        from ..ingest import BagCacheView, dataset_scope, default_budget, device_bag_cache, new_scope_token
        caches = self.__dict__.setdefault("_bag_caches", {})
        scope = dataset_scope(loader) if loader is not None else None
        gb = os.environ.get("ADVMIL_BAG_CACHE_GB", self.cfg.get("bag_cache_gb"))
        if scope is None:
            # An iterable without a dataset object (a list, a generator) has no identity of its own: its bags are kept -- under
            # (this handler, name_loader) -- only when the caller asked for the cache explicitly (cfg['bag_cache_gb'] /
            # ADVMIL_BAG_CACHE_GB); by default such a loader is re-read every epoch, as the reference does (model_handler.py:315).
            scope = ("h", self.__dict__.setdefault("_cache_token", new_scope_token()), name_loader) if gb is not None else False
        scopes = self.__dict__.setdefault("_bag_cache_scopes", {})
        if name_loader not in caches or scopes.get(name_loader) != scope:
            view = None
            if scope is not False and (gb is None or float(gb) > 0):
                cache = device_bag_cache(self.device, default_budget(self.device) if gb is None else float(gb) * 1e9)
                view = BagCacheView(cache, scope)
                if scope[0] == "h":                  # bags scoped by this handler leave the device's cache with it
                    import weakref
                    weakref.finalize(self, cache.drop_scope, scope)
            caches[name_loader], scopes[name_loader] = view, scope
        return caches[name_loader]

    # ------------------------------------------------------------------------------------------
    def _update_disc(self, i_batch, xs, ys, mode="wlabel", label_visible_mask=None, ys_host=None, noise=None, plan=None,
                     defer_apply=False):
        """netD.train(), netG.eval(); real pairs only for event bags with a visible label, fake pairs for all.
        `noise`: optional per-bag injected generator noise (tests). `plan`: the step plan (`_plan`) when the caller already built
        it for this step batch. `defer_apply`: only start D's gradient exchange; the next `_update_gen` completes the D update
        (reduce wait, log, Adam) behind its backbone forward. Returns (pred_collector, fake_collector)."""
        if plan is None:
            plan = self._plan(xs, ys, mode, label_visible_mask, ys_host)
        preds, fakes = self._disc_backward(i_batch, xs, ys, plan, noise)
        if defer_apply:
            self._pending_d = self.dp.allreduce_async(self.optimizerD.flat_grad, self._st_d[0])
        else:
            self._disc_apply()
        return preds, fakes

    def _plan(self, xs, ys, mode, label_visible_mask, ys_host, y_stack=None, pad=0):
        """Host-side facts of a step batch (built OUTSIDE HIP-graph capture): which bags feed a real pair / the supervised loss,
        the GLOBAL denominators of the reference's means, the row segments of the step slab and -- under bag-parallel -- the maps
        from this rank's rows to the rows of the single-process slab that index every dropout / noise draw. All device arrays are
        assembled on the host and sent with asynchronous copies from pinned memory: no stream synchronisation at world == 1, one
        tiny all-gather (counts + bag lengths) at world > 1."""
        import numpy as np
        n = len(xs)
        dev = self.device
        vis = self._vis(mode, n, label_visible_mask)
        if ys_host is None:
            ys_host = [y.cpu() for y in ys]             # fallback: one sync (the epoch loop passes host labels)
        is_real = [bool(float(yh[0, 1]) == 1.0) and vis[i] for i, yh in enumerate(ys_host)]
        lens = [self._rows(x[0]) for x in xs]
        W, r = self.dp.world, self.dp.rank
        counts = [sum(is_real), n, sum(vis)]
        rng_rows = rowoff16 = None
        if W > 1:
            allv = self.dp.allgather_ints(counts + [int(pad)] + lens, dev)
            n_real, n_fake, n_vis = (sum(v[k] for v in allv) for k in range(3))
            rng_rows, rowoff16 = self._rng_row_maps([v[4:] for v in allv], lens, n, W, r, [v[3] for v in allv])
        else:
            n_real, n_fake, n_vis = counts
            if pad:                                      # the real rows draw what they draw in the unpadded slab (parallel.rng_row_maps)
                rng_rows, rowoff16 = self._rng_row_maps([lens], lens, n, 1, 0, [int(pad)])
        masks = torch.empty(2 * n, dtype=torch.float32, pin_memory=True)
        mv = masks.numpy()
        mv[:n] = [1.0 if q else 0.0 for q in is_real]
        mv[n:] = [1.0 if v else 0.0 for v in vis]
        masks_d = masks.to(dev, non_blocking=True)
        # `pad` zero rows behind the bags (SlabStager.pad_rows) are one more segment of the slab: a dummy bag whose pooled row is
        # dropped (`_bags`) before anything bag-level sees it
        seg = ops.Segments(lens + [pad] if pad else lens, dev)
        sel2 = None
        if pad:                                          # rows of the D update's stacked [fake | real] pooled features that are bags
            sel2 = torch.tensor(list(range(n)) + list(range(n + 1, 2 * n + 1)), dtype=torch.int64).pin_memory().to(dev, non_blocking=True)
        seg16 = seg.div(16)                              # D's region embedding needs N % 16 == 0 (backbone_utils.py:65)
        seg16.twice()                                    # (built here: the D update stacks its fake and real passes)
        seg16.rng_rowoff = rowoff16
        self._plan_count = getattr(self, "_plan_count", 0) + 1
        y = torch.cat(ys, dim=0) if y_stack is None else y_stack
        # the D update's stacked label column [fake predictions | real labels]: the lower half is constant per plan, the upper half is
        # written by the generator's head itself (no concatenation launch per step)
        t2 = torch.cat([torch.zeros(n, 1, dtype=torch.float32, device=dev), y[:, 0:1].float()], dim=0) if n_real > 0 else None
        return SimpleNamespace(vis=vis, is_real=is_real, n_real=n_real, n_fake=n_fake, n_vis=n_vis, y=y, t2=t2,
                               y_t=y[:, 0:1].contiguous(), y_e=y[:, 1:2].contiguous(),     # label columns, contiguous once per plan
                               vis_mask=None if all(vis) else masks_d[n:], real_mask=masks_d[:n], seg=seg, seg16=seg16,
                               rng_rows=rng_rows, token=self._plan_count, _keep=(masks, masks_d), _X=None, _X_src=None,
                               pad=int(pad), nb=n, sel2=sel2)

    def _rng_row_maps(self, all_lens, lens, n, W, r, pads=None):
        """Bag-parallel: upload parallel.rng_row_maps (local row -> row in the single-process slab, one map per row LAYOUT a
        slab-level tensor of the step can have) as ONE pinned buffer / one asynchronous copy -> (ops.DeviceRng.rows keyed by layout
        kind, Segments.rng_rowoff)."""
        from ..parallel import rng_row_maps
        ident = {}
        if W == 1:
            # single process with a pad: every layout but the stacked one is the identity on its real rows and the pad rows sit behind
            # them (their draws are never used) -> no lookup in the kernels, no map to build; the attention offsets of the real bags
            # are zero. Same indices as parallel.rng_row_maps(..., pads) gives for W = 1 (tests/test_parallel_gloo_cpu.py).
            import numpy as np
            pad = int(pads[0])
            L, p16 = sum(lens) // 16, (pad + 15) // 16
            ar, ap = np.arange(L, dtype=np.int64), np.arange(p16, dtype=np.int64)
            maps = {"region2": np.concatenate([ar, 2 * L + ap, L + ar, 2 * L + p16 + 1 + ap])}
            rows_of = {"patch": sum(lens) + pad, "region": L + p16, "bag": n, "bag2": 2 * n, "cluster": 8 * (n + 1)}
            ident = {k: ops.IdentityRows(v) for k, v in rows_of.items() if k != "cluster" or self.bcb == "cluster"}
            off16 = np.zeros(0, dtype=np.int64)
        else:
            maps, off16 = rng_row_maps(all_lens, W, r, cluster=self.bcb == "cluster", pads=pads)
        kinds = list(maps)
        n = int(off16.shape[0])                          # (one more entry when this rank's slab carries a pad bag)
        host = torch.empty(sum(int(maps[k].shape[0]) for k in kinds) + n, dtype=torch.int64, pin_memory=True)
        hv, o, spans = host.numpy(), 0, {}
        for k in kinds:
            m = int(maps[k].shape[0])
            hv[o:o + m] = maps[k]; spans[k] = (o, o + m); o += m
        hv[o:o + n] = off16
        devbuf = host.to(self.device, non_blocking=True)
        self._rng_keep = (host, devbuf)
        out = {k: devbuf[a_:b_] for k, (a_, b_) in spans.items()}
        out.update(ident)
        return out, (None if W == 1 else devbuf[o:o + n])

    @staticmethod
    def _rows(x):
        return x.shape[-2]

    def _slab(self, xs, plan=None):
        """The step's bags as ONE [N_total, C] matrix: a zero-copy view when they already sit back to back in one
        allocation (the staging slab of the loader / the resident pool of the bench), else one concatenation (kept on the step
        plan: the D and the G update of a step read the same matrix, and the forward memo recognises it by its address)."""
        if plan is not None and getattr(plan, "_X", None) is not None and plan._X_src == [id(x[0]) for x in xs]:
            return plan._X
        X = self._slab_build(xs, 0 if plan is None else getattr(plan, "pad", 0))
        if plan is not None:
            plan._X, plan._X_src = X, [id(x[0]) for x in xs]
        return X

    def _slab_build(self, xs, pad=0):
        return MyHandler._slab_build_static(xs, getattr(self, "resident_planes", True), pad)

    @staticmethod
    def _bags(t, plan, stacked=False):
        """Drop the dummy bag of the slab pad (`_plan`) from a pooled [B + 1, d] tensor ([2 (B + 1), d] when `stacked`)."""
        if t is None or not getattr(plan, "pad", 0):
            return t
        return t.index_select(0, plan.sel2) if stacked else t[:plan.nb]

    @staticmethod
    def _slab_build_static(xs, resident_planes=True, pad=0):
        x0 = xs[0][0]
        c = x0.shape[-1]
        rows = [x[0].shape[-2] for x in xs]
        ok = all(x[0].is_contiguous() for x in xs)
        if ok:
            st = x0.untyped_storage().data_ptr()
            ptr = x0.data_ptr()
            for x, r in zip(xs, rows):
                if x[0].data_ptr() != ptr or x[0].untyped_storage().data_ptr() != st:
                    ok = False
                    break
                ptr += r * c * x0.element_size()
        if pad and not ok:
            raise RuntimeError("a slab pad needs the bags back to back in the staging slab")
        X = (x0.as_strided((sum(rows) + pad, c), (c, 1), x0.storage_offset()) if ok       # (pad: zero rows the stager put behind them)
             else torch.cat([x[0].reshape(-1, c) for x in xs], dim=0))
        if ops.is_bf16_slab(X):
            return X                             # x_storage = 'bf16': the slab is its own (single) operand plane, nothing to derive
        spl = getattr(x0, "_advmil_stager_planes", None) if ok else None
        stale = bool(getattr(x0, "_advmil_fp32_stale", False))
        if spl is not None and spl.hi.shape[0] == X.shape[0] and ops.slab_takes_planes(X.shape[0], c):
            X._advmil_planes = spl               # assembled by the staging slab from the cached bags' planes (copy stream)
            if stale:
                # cached bags were staged as planes only (ingest.SlabStager.add_device): nothing may read this slab's fp32 rows
                X._advmil_planes = ops.Planes(spl.hi, spl.lo)
                X._advmil_planes.fp32_stale = True
                X._advmil_fp32_stale = True
            return X
        if stale:
            raise RuntimeError("advmil_amd: a step slab staged as operand planes only cannot be read as fp32 rows (planes missing / not "
                               "back to back / arithmetic mode changed between staging and the step)")
        pls = None if ok else [getattr(x[0], "_advmil_bag_planes", None) for x in xs]
        if (pls and all(p is not None for p in pls) and X.shape[0] >= 4096 and ops.USE_PLANES and ops.get_gemm_mode() == "bf16x3"
                and ops.gemm_plan_planes(X.shape[0], 128, c)):
            # bags from the device-resident cache carry their operand planes: the step's planes are a row gather, not a re-split
            xpl = ops.Planes.alloc(tuple(X.shape), X.device)          # (one allocation for both planes: ops.Planes.alloc)
            torch.cat([p.hi for p in pls], dim=0, out=xpl.hi)
            torch.cat([p.lo for p in pls], dim=0, out=xpl.lo)
            X._advmil_planes = xpl
        else:
            MyHandler._slab_planes_static(X, x0 if ok else None, resident_planes)
        return X

    def _slab_planes(self, X, anchor):
        MyHandler._slab_planes_static(X, anchor, getattr(self, "resident_planes", True))

    @staticmethod
    def _slab_planes_static(X, anchor, resident_planes=True):
        """bf16x3 mode: the slab's operand planes (hi = bf16(x), lo = bf16(x - hi); the same 4 bytes per element as the fp32 rows).
        Every contraction that reads X (the generator's and the discriminator's embedding FCs, twice per step each) then takes the
        plane-fed LDS-DMA kernel instead of re-splitting the rows in every workgroup. A resident slab (zero-copy view starting at
        the first bag's tensor `anchor`: the loader's staging buffer / the bench's pool) is split ONCE: the planes are kept on the
        anchor tensor object and reused until its storage is written again (version counter); a slab assembled by a copy
        (anchor None) is split per step."""
        if X.shape[0] < 4096 or not ops.USE_PLANES or ops.get_gemm_mode() != "bf16x3":
            return
        if not ops.SLAB_PLANES_ANY and not ops.gemm_plan_planes(X.shape[0], 128, X.shape[1]):
            return                            # (until round 5: only slabs the plane-fed NT kernel's 256-row tiles fill the chip with)
        if anchor is None or not resident_planes:
            # (resident_planes = False: fp32-only residency -- the split is part of every step, also under HIP-graph replay)
            X._advmil_planes = ops.split_planes(X)
            return
        cache = anchor.__dict__.setdefault("_advmil_slab_planes", {})
        ent = cache.get(X.shape[0])
        if ent is None or ent[0] != X._version:
            ent = (X._version, ops.split_planes(X, out=None if ent is None else ent[1]))
            cache[X.shape[0]] = ent
        X._advmil_planes = ent[1]

    @staticmethod
    def _stack_noise(noise):
        """per-bag [[n_layer0, ...], ...] -> per-layer [B, w] stacks for the batched head."""
        if noise is None:
            return None
        return [torch.cat([nb[j] for nb in noise], dim=0) for j in range(len(noise[0]))]

    def _prefill_first_layers(self, X):
        """The generator's eval forward and the discriminator's forward of the D update both start with a Linear over the step
        slab X (model/backbone.py:60-66 / backbone_utils.py:158-168; model_utils.py:130-140): ops.prefill_two_layers runs the two
        as one plane-fed launch. Layer lookup is by backbone kind; anything else simply takes the ordinary path."""
        return MyHandler._prefill_static(self.netG, self.netD, self.bcb, X)

    @staticmethod
    def _prefill_static(netG, netD, bcb, X):
        bb = netG.backbone
        if bcb == "abmil" and hasattr(bb, "attention_net"):
            fc = bb.attention_net[0]
            l1 = (fc.weight, fc.bias, "relu", True)
        elif bcb == "patch" and hasattr(bb, "patch_embedding_layer") and hasattr(bb.patch_embedding_layer, "conv"):
            cv = bb.patch_embedding_layer.conv
            l1 = (cv.weight, cv.bias, "none", False)
        else:
            return False
        emb = getattr(getattr(netD, "net_pair_one", None), "embedding", None)
        if emb is None or not hasattr(emb, "conv"):
            return False
        return ops.prefill_two_layers(X, l1, (emb.conv.weight, emb.conv.bias, "none", False))

    def _gen_features(self, X, plan, xs):
        """Generator backbone over the whole step slab -> [B, d]. `patch` mode skips coords (model_handler.py:390)."""
        exts = [x[1] for x in xs] if self.bcb in ("cluster", "graph") else None
        pad = getattr(plan, "pad", 0)
        if self.bcb == "cluster" and all(torch.is_tensor(e) and not e.is_cuda for e in exts):
            # host ids (the epoch loop leaves them there): the step's ids -- and zeros for the pad's dummy bag -- as one pinned buffer,
            # one asynchronous copy per step plan instead of one synchronous pageable copy per bag
            ids = plan.__dict__.get("_cluster_ids")
            if ids is None:
                n = sum(int(e.numel()) for e in exts)
                buf = torch.zeros(n + pad, dtype=exts[0].dtype, pin_memory=True)
                torch.cat([e.reshape(-1) for e in exts], out=buf[:n])
                ids = plan._cluster_ids = buf.to(self.device, non_blocking=True)
            exts = [ids]
        elif exts is not None and pad:
            exts = exts + [torch.zeros(pad, dtype=exts[0].dtype, device=exts[0].device)]          # the dummy bag's cluster ids
        return self._bags(self.netG.features_multi(X, plan.seg, exts), plan)

    def _disc_backward(self, i_batch, xs, ys, plan, noise=None):
        """Capturable (no host sync, no collective): zero D grads, forward of the step slab, ONE backward of the D loss.
        All N-row / region-row kernels run once over the B bags' rows (segmented softmax-pool per bag); the [1,d]-sized
        heads and tails run once on [B,d] stacks."""
        self._set_mode(self.netD, True)
        self._set_mode(self.netG, False)
        self.rng.rows = plan.rng_rows          # (cleared again below: the process-wide rng must not carry this step's maps)
        try:
            return self._disc_backward_body(i_batch, xs, ys, plan, noise)
        finally:
            self.rng.rows = None

    def _disc_backward_body(self, i_batch, xs, ys, plan, noise=None):
        self.optimizerD.zero_grad()
        X = self._slab(xs, plan)
        y = plan.y                           # [B, 2] label stack, built once per step plan
        self._prefill_first_layers(X)          # G's and D's first layers over the slab from one launch (X staged once), when possible
        ops.MEMO.begin("record", ("G", id(self.netG), getattr(self.optimizerG, "n_updates", 0), plan.token), X)
        try:
            with torch.no_grad():                                              # the reference builds, then detaches (400)
                pred = self.netG.finish(self._gen_features(X, plan, xs), noise=self._stack_noise(noise),
                                        pred_out=None if plan.t2 is None else plan.t2[:len(xs)])               # [B,1]
        finally:
            ops.MEMO.end()
        f_real = f2 = None
        # the region embedding is shared by the real and the fake pairs; with real pairs in the step it leaves its kernel stacked twice
        emb = self.netD.embed_rows(X, 2) if plan.n_real > 0 else self.netD.embed_rows(X)
        if plan.n_real > 0:                  # GLOBAL count: every rank of a bag-parallel step takes the same branch / draws
            # The fake pairs (all bags) and the real pairs go through the region-level network and the tail as ONE stacked batch:
            # rows [0, L) are the fake pass, rows [L, 2L) the real pass (its own dropout draw, as a separate forward has, because
            # the draw is indexed by row). Every kernel of the tail -- ~120 launches per pass -- runs once instead of twice; real
            # scores of bags without a visible event are computed and dropped (B rows of [B,d] work).
            nb = len(xs)
            eb2, im2 = self.netD.bag_features_multi(emb, plan.seg16.twice())
            eb2, im2 = self._bags(eb2, plan, True), self._bags(im2, plan, True)
            # (pred IS the upper half of plan.t2 when the head wrote it there)
            t2 = plan.t2 if (plan.t2 is not None and pred.data_ptr() == plan.t2.data_ptr()) else torch.cat([pred, plan.y_t], dim=0)
            f2 = self.netD.tail(eb2, im2, t2).view(-1)
            f_fake = f2.detach()[:nb]                                           # (the loss takes f2 whole: no slice backward)
        else:
            eb, im = self.netD.bag_features_multi(emb, plan.seg16)
            f_fake = self.netD.tail(self._bags(eb, plan), self._bags(im, plan), pred).view(-1)
        # real_fake_loss with the global denominators (loss/utils.py:182-203, model_handler.py:412) as ONE launch that also yields
        # d loss / d score; the real pairs are selected by a 0/1 mask (same sum as f_real[event & visible], no index backward)
        ops.PREFILL.clear()                  # (anything the two forwards did not take is stale from here on)
        ops.DY_PLANES.clear()
        if f2 is not None:                   # rows [0, nb) fake, [nb, 2 nb) real scores of ALL bags; the mask picks the real pairs
            loss, st = ops.gan_d_loss_stacked(f2, len(xs), plan.real_mask, self.which_loss, plan.n_fake, plan.n_real, root=True)
        else:
            loss, st = ops.gan_d_loss(f_fake, None, None, self.which_loss, plan.n_fake, plan.n_real, root=True)
        # (deferred_sums: the backward's ~7 merge launches of parameter-gradient partials become one, issued on exit)
        try:
            with torch.autograd.set_multithreading_enabled(False), ops.deferred_sums():       # backward on THIS thread (see _gen_finish)
                torch.autograd.backward(loss, grad_tensors=self._one())  # (the root gradient is a cached 1: no fill launch per step)
        finally:
            ops.DY_PLANES.clear()            # (a backward that raised may have left a plane hand-over behind: never let it meet a reused address)
        self._st_d = (st, plan, i_batch)     # this rank's partial sums over the global denominators; reduced + logged in _disc_apply
        if plan.t2 is not None and pred.data_ptr() == plan.t2.data_ptr() and not torch.cuda.is_current_stream_capturing():
            pred = pred.clone()              # the plan's buffer is rewritten by its next step; the collector keeps these
        preds = list(pred.split(1, dim=0))
        fakes = list(f_fake.detach().split(1, dim=0))
        return preds, fakes

    def _one(self):
        return self._one_t               # made in __init__: a first use inside HIP-graph capture would put it into the graph's pool

    def _reduce_d(self):
        """Bag-parallel exchange of the D update: the flat gradient arena and the step's three loss statistics."""
        self.dp.allreduce_(self.optimizerD.flat_grad)
        self.dp.allreduce_(self._st_d[0])

    def _log_d(self):
        st, plan, i_batch = self._st_d
        nr, nf = max(plan.n_real, 1), plan.n_fake
        self.log(LazyLog((st,), lambda v: {"train_batch/netD/Loss_D": v[0], "train_batch/netD/D_real": v[1] / nr,
                                           "train_batch/netD/D_fake": v[2] / nf, "i_batch": i_batch}))

    def _disc_apply(self):
        self._reduce_d()
        self._log_d()
        self.optimizerD.step()

    def _disc_finish_deferred(self):
        """Complete a D update whose exchange `_update_disc(defer_apply=True)` started."""
        pend = self.__dict__.pop("_pending_d", None)
        if pend is None:
            return
        for w in pend:
            w.wait()
        self._log_d()
        self.optimizerD.step()

    # ------------------------------------------------------------------------------------------
    def _update_gen(self, i_batch, xs, ys, mode="wlabel", label_visible_mask=None, ys_host=None, noise=None, plan=None):
        """netD.eval(), netG.train(); gen_total = t_reg + coef * (-mean f_fake) + l1 * sum|W_G|."""
        if plan is None:
            plan = self._plan(xs, ys, mode, label_visible_mask, ys_host)
        self._gen_forward(xs, plan, noise)
        self._disc_finish_deferred()         # D's exchange ran under the generator's backbone forward; D must be stepped before it scores
        self._gen_finish(i_batch, xs, ys, plan)
        self._gen_apply()

    def _gen_backward(self, i_batch, xs, ys, plan, noise=None):
        """Capturable: zero G grads, forward of the step slab, ONE backward of the G loss."""
        self._gen_forward(xs, plan, noise)
        self._gen_finish(i_batch, xs, ys, plan)

    def _gen_forward(self, xs, plan, noise=None):
        """The part of the generator update that does not depend on D: zero G grads, backbone + head forward (graph kept)."""
        self._set_mode(self.netD, False)
        self._set_mode(self.netG, True)
        self.rng.rows = plan.rng_rows
        try:
            self._gen_forward_body(xs, plan, noise)
        finally:
            self.rng.rows = None

    def _gen_forward_body(self, xs, plan, noise=None):
        self.optimizerG.zero_grad()
        X = self._slab(xs, plan)
        # row-sized pre-dropout layer outputs of the eval forward in _disc_backward are reused -- only when this call belongs
        # to the SAME step plan (token) and the generator has not been updated since (n_updates); an unpaired call recomputes
        ops.MEMO.begin("replay", ("G", id(self.netG), getattr(self.optimizerG, "n_updates", 0), plan.token), X)
        try:
            feats = self._gen_features(X, plan, xs)
        finally:
            ops.MEMO.end(clear=True)
        pred = self.netG.finish(feats, noise=self._stack_noise(noise))               # pred [B,1], graph kept
        self._g_fwd = (X, pred)

    def _gen_finish(self, i_batch, xs, ys, plan):
        """D's score of the predictions (updated D, frozen), the G loss and its ONE backward."""
        X, pred = self.__dict__.pop("_g_fwd")
        with torch.no_grad():                                                  # nothing of D(x) depends on G
            eb, im = self.netD.bag_features_multi(self.netD.embed_rows(X), plan.seg16)
            eb, im = self._bags(eb, plan), self._bags(im, plan)
        # The generator loss only needs d f / d pred. The reference lets autograd also fill netD's weight gradients here and
        # throws them away at the next optimizerD.zero_grad() (model_handler.py:409, 497); with D's parameters frozen for this
        # backward those contractions (five small dW launches + their reductions) are never issued.
        d_params = [p for p in self.netD.parameters() if p.requires_grad]
        for p in d_params:
            p.requires_grad_(False)
        try:
            f_fake = self.netD.tail(eb, im, pred).view(-1)
        finally:
            for p in d_params:
                p.requires_grad_(True)
        # gen_total = recon_loss over the visible labels + coef * (-mean f_fake) (model_handler.py:468-486) as ONE launch that also
        # yields the gradients w.r.t. pred and f_fake
        y = plan.y
        n_vis = plan.n_vis if (plan.n_vis > 0 and any(plan.vis)) else 0
        if self.dp.world > 1:
            n_vis = plan.n_vis
        rc = self._recon
        total, st = ops.gan_g_loss(pred, f_fake, plan.y_t, plan.y_e, plan.vis_mask, rc["alpha"], rc["gamma"], rc["norm"],
                                   self.coef_ganloss, plan.n_fake, n_vis, root=True)
        # The engine would hand a CUDA graph to its device thread; every node of ours is a short Python function that only enqueues
        # launches, so the hand-over and the GIL ping-pong cost more than they buy (host issue per eager step 2.9 -> 2.5 ms,
        # tools/probe/eager_host_profile.py). Scoped: the caller's setting comes back on exit.
        try:
            with torch.autograd.set_multithreading_enabled(False), ops.deferred_sums():
                torch.autograd.backward(total, grad_tensors=self._one())
        finally:
            ops.DY_PLANES.clear()
        self._st_g = (st, i_batch)

    def _reduce_g(self):
        self.dp.allreduce_(self.optimizerG.flat_grad)
        self.dp.allreduce_(self._st_g[0])

    def _log_g(self):
        st, i_batch = self._st_g
        c1 = self.coef_l1 if self.coef_l1 > 1e-8 else 0.0      # added once, after the reduce: the L1 term is not a per-bag sum
        # sum |W| of the L1 term's logged value: per-workgroup shares written by the Adam launch that follows (the parameters as they
        # stand before the update), summed on the host when the log is read -- no pass of its own over the arena
        self._abs_partial = torch.empty(ops.adam_blocks(self.optimizerG.flat_param.numel()), dtype=torch.float32, device=self.device) if c1 else None
        tens = (st, self._abs_partial) if c1 else (st,)
        self.log(LazyLog(tens, lambda v, a=None: {"train_batch/netG/Loss_G_fake": v[2], "train_batch/netG/Loss_G_time": v[1],
                                                  "train_batch/netG/Loss_G_total": v[0] + (c1 * a.sum(dtype=a.dtype) if a is not None else 0.0),
                                                  "train_batch/netG/D_fake_avg": -v[2], "i_batch": i_batch}))

    def _gen_apply(self):
        self._reduce_g()
        self._log_g()
        self.optimizerG.step(abs_partial=self._abs_partial)                    # L1 sub-gradient folded in

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def test_model(modelG, modelD, backbone, loader, times_test_sample=1, checkpoints=None, test_zero_noise=False, noise=None,
                   batch_bags=None, x_storage=None):
        """Eval: y_hat, f_fake, and `times_test_sample` more generator samples + their median per bag (reference
        model_handler.py:598-643: one synchronous `.cuda()`, 1 + times_test_sample full generator forwards, one D forward and
        4-6 `.cpu()` syncs PER BAG -- and `_run_training` runs it over the validation and the test set after every epoch).
        Here: in eval mode the backbone output is identical across the samples (only the head noise differs), so a bag is embedded
        ONCE and the head sampled times_test_sample + 1 times; and `batch_bags` bags (default 16, ADVMIL_EVAL_BATCH_BAGS) go
        through the generator and the discriminator as ONE step slab (the training step's slab kernels), staged through the pinned
        double-buffered slab on the copy stream and kept in the device-resident bag cache across epochs (keyed by the loader's
        dataset object and patient index, shared budget with the training loop). The collector comes back with one D2H per key.
        PatchGCN bags, bags whose length is not a multiple of 16 and injected-noise calls with ragged lists take the per-bag path.
        `noise`: optional per-bag list of injected noise tensors (tests)."""
        if checkpoints is not None:
            dev = next(modelG.parameters()).device
            modelG.load_state_dict(torch.load(checkpoints[0], map_location=dev)["model"])
            modelD.load_state_dict(torch.load(checkpoints[1], map_location=dev)["model"])
        modelG.eval(); modelD.eval()
        dev = next(modelG.parameters()).device
        nb_max = int(batch_bags if batch_bags is not None else os.environ.get("ADVMIL_EVAL_BATCH_BAGS", "16"))
        keys = ["idx", "y", "y_hat", "f_fake"] + (["dist_y_hat", "avg_y_hat"] if times_test_sample > 1 else [])
        parts = {k: [] for k in keys}                 # device (or host, for idx / y) pieces in loader order; ONE .cpu() per key at the end

        def one_bag(b, idx, x, y):
            x_data, x_ext = [t.to(dev) if torch.is_tensor(t) else t for t in x]
            if backbone == "graph":
                H = modelG.backbone(x_ext, None)
            elif backbone == "patch":
                H = modelG.backbone(x_data, None)
            else:
                H = modelG.backbone(x_data, x_ext)
            it = iter(noise[b]) if noise is not None else None
            y_hat = modelG.head(H, test_zero_noise, None if it is None else [next(it)])
            parts["idx"].append(idx.detach()); parts["y"].append(y.detach())
            parts["y_hat"].append(y_hat); parts["f_fake"].append(modelD(x_data, y_hat))
            if times_test_sample > 1:
                ys = torch.stack([modelG.head(H, test_zero_noise, None if it is None else [next(it)]) for _ in range(times_test_sample)])
                parts["dist_y_hat"].append(ys.transpose(0, 1)); parts["avg_y_hat"].append(torch.median(ys, dim=0)[0])

        def slab_batch(bt):
            """One StepBatch (ingest.step_batches) through the slab kernels."""
            xs, n, pad = bt.xs, len(bt.xs), bt.pad
            X = MyHandler._slab_build_static(xs, True, pad)
            lens = [x[0].shape[-2] for x in xs]
            seg = ops.Segments(lens + [pad] if pad else lens, dev)       # (the pad's zero rows are a dummy bag, dropped below)
            MyHandler._prefill_static(modelG, modelD, backbone, X)       # G's and D's first layers over the slab from one launch
            exts = [x[1].to(dev) if torch.is_tensor(x[1]) else x[1] for x in xs] if backbone == "cluster" else None
            if exts is not None and pad:
                exts = exts + [torch.zeros(pad, dtype=exts[0].dtype, device=dev)]
            feats = modelG.features_multi(X, seg, exts)[:n]                # [B, d]: the bags are embedded ONCE
            bb = modelG.backbone
            H = bb.post(feats) if hasattr(bb, "post") else feats
            nz = None
            if noise is not None:                                          # per-bag [n0, n1, ...] -> per-call [B, w] stacks
                nz = [torch.cat([noise[b][c] for b in bt.pos], dim=0) for c in range(len(noise[bt.pos[0]]))]
            y_hat = modelG.head(H, test_zero_noise, None if nz is None else [nz[0]])
            emb = modelD.embed_rows(X)
            eb, im = modelD.bag_features_multi(emb, seg.div(16))
            f_fake = modelD.tail(eb[:n], None if im is None else im[:n], y_hat)
            ops.PREFILL.clear()
            ops.DY_PLANES.clear()
            parts["idx"].extend(i.detach() for i in bt.idx); parts["y"].extend(y.detach() for y in bt.ys)
            parts["y_hat"].append(y_hat); parts["f_fake"].append(f_fake.reshape(n, -1))
            if times_test_sample > 1:
                ys = torch.stack([modelG.head(H, test_zero_noise, None if nz is None else [nz[1 + c]]) for c in range(times_test_sample)])
                parts["dist_y_hat"].append(ys.transpose(0, 1)); parts["avg_y_hat"].append(torch.median(ys, dim=0)[0])

        from ..ingest import loader_cache_view, step_batches
        slab_able = (backbone != "graph" and hasattr(modelG, "features_multi") and hasattr(modelD, "bag_features_multi")
                     and (noise is None or len({len(nb) for nb in noise}) == 1))
        if x_storage is None:                # (the handler that built modelG stamped its storage mode on it; else ADVMIL_X_STORAGE / fp32)
            x_storage = getattr(modelG, "_advmil_x_storage", None)
        with torch.no_grad():
            for bt in step_batches(loader, dev, nb_max if slab_able else 1, loader_cache_view(dev, loader),
                                   stageable=lambda x0: x0.shape[1] % 16 == 0, pad_multiple=int(os.environ.get("ADVMIL_SLAB_PAD", "256")),
                                   x_storage=x_storage):
                if bt.staged:
                    slab_batch(bt)
                else:
                    one_bag(bt.pos[0], bt.idx[0], bt.xs[0], bt.ys[0])
        res = {k: None for k in ("idx", "y", "y_hat", "f_fake")}
        if parts["idx"]:
            res = {k: torch.cat([t if t.dim() > 0 else t.reshape(1) for t in parts[k]], dim=0).detach().cpu() for k in keys}
        return res

    # ------------------------------------------------------------------------------------------
    def _get_state_dict(self, epoch, model="G"):
        net, opt = (self.netG, self.optimizerG) if model == "G" else (self.netD, self.optimizerD)
        return {"epoch": epoch, "model": net.state_dict(), "optimizer": opt.state_dict()}

    @staticmethod
    def _prefixed(path, prefix):
        d, f = osp.split(path)
        return osp.join(d, prefix + "_" + f)

    def save_model(self, epoch, ckpt_type="best", run_name="train"):
        if ckpt_type not in ("best", "last"):
            raise KeyError("Expected best or last for `ckpt_type`, but got {}.".format(ckpt_type))
        pg = self.last_netG_ckpt_path if ckpt_type == "last" else self.best_netG_ckpt_path
        pd = self.last_netD_ckpt_path if ckpt_type == "last" else self.best_netD_ckpt_path
        torch.save(self._get_state_dict(epoch, "G"), self._prefixed(pg, run_name))
        torch.save(self._get_state_dict(epoch, "D"), self._prefixed(pd, run_name))

    def resume_model(self, ckpt_type="best", run_name="train"):
        if ckpt_type not in ("best", "last"):
            raise KeyError("Expected best or last for `ckpt_type`, but got {}.".format(ckpt_type))
        pg = self.last_netG_ckpt_path if ckpt_type == "last" else self.best_netG_ckpt_path
        pd = self.last_netD_ckpt_path if ckpt_type == "last" else self.best_netD_ckpt_path
        g = torch.load(self._prefixed(pg, run_name), map_location=self.device)
        d = torch.load(self._prefixed(pd, run_name), map_location=self.device)
        self.netG.load_state_dict(g["model"]); self.optimizerG.load_state_dict(g["optimizer"])
        self.netD.load_state_dict(d["model"]); self.optimizerD.load_state_dict(d["optimizer"])
        ops.MEMO.end(clear=True)             # nothing recorded under the old weights may be replayed

    # ---- orchestration around the step is the reference's own (dataset / evaluator / wandb): INTEGRATION.md
    def exec(self):
        raise NotImplementedError("dataset / evaluator orchestration is out of scope here; see INTEGRATION.md for "
                                  "binding this class under the reference's main.py")

    exec_test = exec_semi_sl = exec
