"""Optimizer for the G+D step: torch.optim.Adam semantics (L2-in-grad weight decay, the reference's
`add_weight_decay` filter: no decay on 1-D tensors and *.bias -- optim/optim_factory.py:25-37,76-77;
D: model/model_handler.py:107) executed as ONE fused HIP launch over a flat fp32 parameter arena.

Parameters, gradients and both Adam moments live in four contiguous buffers; every nn.Parameter is a
view into the arena, every .grad a view into the gradient arena (so a bag-parallel step all-reduces one
tensor per network). The L1 regulariser of loss/utils.py:6-14 is applied inside the same kernel as
coef*sign(w). state_dict() has torch.optim.Adam's layout, so reference checkpoints resume."""
import torch

from . import ops


def create_optimizer(args, model, filter_bias_and_bn=True):
    """Reference signature (optim/optim_factory.py:40). Only `adam` is reachable from cfg_nlst.yaml:63."""
    if args.opt.lower().split("_")[-1] != "adam":
        raise NotImplementedError(f"opt_netG={args.opt}: the AdvMIL configs use adam")
    wd = args.weight_decay or 0.0
    kw = {}
    if getattr(args, "opt_eps", None) is not None:
        kw["eps"] = args.opt_eps
    if getattr(args, "opt_betas", None) is not None:
        kw["betas"] = args.opt_betas
    return FlatAdam(model, lr=args.lr, weight_decay=wd, filter_bias_and_bn=bool(wd and filter_bias_and_bn), **kw)


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, filter_bias_and_bn=False, l1_coef=0.0):
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        if not named:
            raise ValueError("no parameters")
        dev = named[0][1].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdam runs the fused HIP Adam kernel: move the model to the GPU first")
        if filter_bias_and_bn:
            no_decay = [(n, p) for n, p in named if p.dim() == 1 or n.endswith(".bias")]
            decay = [(n, p) for n, p in named if not (p.dim() == 1 or n.endswith(".bias"))]
            groups = [{"params": [p for _, p in no_decay], "weight_decay": 0.0},
                      {"params": [p for _, p in decay], "weight_decay": weight_decay}]
        else:
            groups = [{"params": [p for _, p in named], "weight_decay": weight_decay}]
        # Arena layout (independent of the param_groups / state_dict order above): vectors first, then matrices, each in named
        # order. That puts the two branch weights (and the two branch biases) of every gated-attention scorer side by side, which
        # is what lets the pooling kernels read them as one stacked [2D, D] view and accumulate their gradients in one launch.
        ordered = ([(n, p) for n, p in named if p.dim() == 1 or n.endswith(".bias")]
                   + [(n, p) for n, p in named if not (p.dim() == 1 or n.endswith(".bias"))])
        super().__init__(groups, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.l1_coef = float(l1_coef)
        self.names = [n for n, _ in ordered]
        # ---- arenas (each tensor 8-element aligned: fp32 views are 32 B aligned, the bf16 operand planes' views 16 B aligned)
        offs, total = [], 0
        for _, p in ordered:
            offs.append(total)
            total += (p.numel() + 7) // 8 * 8
        self.flat_param = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        ops.ARENA_STORAGES.add(self.flat_grad.untyped_storage().data_ptr())     # (a deferred split-K fold may only land here: ops.gemm)
        self.flat_m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_wd = torch.zeros(total, dtype=torch.float32, device=dev)
        self.step_t = torch.zeros(1, dtype=torch.int32, device=dev)

        # bf16x3 operand planes of the weights (hi = bf16(w), lo = bf16(w - hi)), same layout as the arena: written by the Adam
        # kernel with every update, so no contraction ever re-splits a weight (ops.weight_planes)
        # (both planes in ONE allocation, lo behind hi: a weight's two planes are then one strided view -- ops.gemm_two_layers stacks the
        # planes of two layers with one launch)
        self.planes = ops.Planes.alloc((total,), dev)
        self.planes.hi.zero_(); self.planes.lo.zero_()
        self._views = []
        with torch.no_grad():
            for (n, p), o in zip(ordered, offs):
                k = p.numel()
                self.flat_param[o:o + k].copy_(p.reshape(-1))
                p.data = self.flat_param[o:o + k].view(p.shape)
                p.grad = self.flat_grad[o:o + k].view(p.shape)
                p._arena_grad = p.grad                     # backward kernels accumulate here directly (ops._arena_grad)
                p._arena_owner = self                      # ... and tell this optimizer that its arena is being written (grad_is_clean)
                self._views.append((p, o, k))
            for g in self.param_groups:
                for p in g["params"]:
                    o, k = next((o, k) for q, o, k in self._views if q is p)
                    self.flat_wd[o:o + k] = g["weight_decay"]
                    self.state[p] = {"step": torch.zeros((), dtype=torch.float32), "exp_avg": self.flat_m[o:o + k].view(p.shape),
                                     "exp_avg_sq": self.flat_v[o:o + k].view(p.shape)}
        self._has_wd = bool(self.flat_wd.abs().max().item() > 0)
        self.refresh_planes()

    def refresh_planes(self):
        """Re-derive the weight planes from the arena (construction, load_state_dict, any torch-side write to a parameter) and
        stamp every parameter with its current version counter."""
        ops.split_planes(self.flat_param, out=self.planes)
        for p, o, k in self._views:
            p._advmil_planes = (self, p._version, ops.Planes(self.planes.hi[o:o + k].view(p.shape), self.planes.lo[o:o + k].view(p.shape)))

    # ---- "is the gradient arena known to be all zero?" A step(clear_grad=True) zeroes the arena behind its read, so the zero_grad()
    # that follows has nothing to fill. The knowledge is HOST state about DEVICE memory, so every writer must be seen:
    #   * kernels that accumulate into arena slots get them from ops._arena_grad, which calls mark_grad_dirty();
    #   * torch-side writes (autograd's AccumulateGrad, p.grad.add_(), ...) move flat_grad's version counter (views share it);
    #   * nothing is recorded while a stream is being CAPTURED (no kernel runs then): a captured step's zero_grad() launches no fill
    #     when the capturer said the graph starts from a clean arena (`capture_assumes_clean`, set by graphed.GraphedStep, whose
    #     replay() makes that true), and the flag is set again by replay() -- never by the captured calls themselves.
    _grad_clean = False
    _clean_version = -1
    capture_assumes_clean = False

    def mark_grad_dirty(self):
        self._grad_clean = False

    def mark_grad_clean(self):
        self._grad_clean, self._clean_version = True, self.flat_grad._version

    def grad_is_clean(self):
        return bool(self._grad_clean and self.flat_grad._version == self._clean_version)

    def zero_grad(self, set_to_none: bool = False):
        if torch.cuda.is_current_stream_capturing():
            if not self.capture_assumes_clean:
                self.flat_grad.zero_()
        elif self.grad_is_clean():
            self._grad_clean = False         # the last step cleared the arena behind its read (step(clear_grad=True)): nothing to fill
        else:
            self.flat_grad.zero_()
            self._grad_clean = False
        for p, o, k in self._views:          # re-attach if someone replaced .grad
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * o:
                p.grad = self.flat_grad[o:o + k].view(p.shape)
                p._arena_grad = p.grad

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0, tick=True, abs_partial=None, clear_grad=False):
        """tick = False: the device step counter is left to the caller (ops.step_seed_tick(self.step_t, seed): the captured step folds it
        into the RNG seed's advance). abs_partial: receives the per-workgroup shares of sum |p| BEFORE the update (the logged L1 term).
        clear_grad: the kernel zeroes the gradient arena behind its read and the next zero_grad() launches nothing -- p.grad then reads
        zero after step(), so only the captured step (nobody looks at gradients between replays) asks for it."""
        g0 = self.param_groups[0]
        b1, b2 = g0["betas"]
        self.n_updates = getattr(self, "n_updates", 0) + 1      # host-side version of the parameters (forward memo key)
        ops.adam_step(self.flat_param, self.flat_grad, self.flat_m, self.flat_v, self.flat_wd if self._has_wd else None,
                      self.step_t, g0["lr"], b1, b2, g0["eps"], grad_scale, self.l1_coef, planes=self.planes, tick=tick,
                      abs_partial=abs_partial, clear_grad=clear_grad)
        if not torch.cuda.is_current_stream_capturing():
            if clear_grad:
                self.mark_grad_clean()
            else:
                self._grad_clean = False

    def state_dict(self):
        n = float(self.step_t.item())
        for st in self.state.values():
            st["step"] = torch.tensor(n)
        return super().state_dict()

    def load_state_dict(self, sd):
        super().load_state_dict(sd)
        step = 0
        with torch.no_grad():
            for p, o, k in self._views:       # pull loaded moments back into the arenas
                st = self.state[p]
                self.flat_m[o:o + k].copy_(st["exp_avg"].reshape(-1))
                self.flat_v[o:o + k].copy_(st["exp_avg_sq"].reshape(-1))
                st["exp_avg"] = self.flat_m[o:o + k].view(p.shape)
                st["exp_avg_sq"] = self.flat_v[o:o + k].view(p.shape)
                step = int(st["step"])
            self.step_t.fill_(step)
            for g in self.param_groups:
                for p in g["params"]:
                    o, k = next((o, k) for q, o, k in self._views if q is p)
                    self.flat_wd[o:o + k] = g["weight_decay"]
        self._has_wd = bool(self.flat_wd.abs().max().item() > 0)
        self.refresh_planes()
