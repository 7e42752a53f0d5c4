"""Flat-arena fused Adam (reference: src/fairseq/optim/fp16_optimizer.py:16-330 +
optim/adam.py:148-228 + utils.py:338-388).

All parameters are re-homed into ONE contiguous low-precision arena (their .data become views), all gradients
into one contiguous gradient arena (their .grad are views, so autograd accumulates in place and the data-parallel
reducer can all-reduce arena slices without packing), and the fp32 master copy / Adam moments live in three more
flat fp32 arrays.  One optimizer step = one sum-of-squares reduction over the gradient arena + one update kernel
(unscale, clip, moments, decoupled weight decay, update, low-precision copy-back), with the clip coefficient
derived on the device: no host synchronisation, 28 B/param of HBM traffic in bf16 mode.
"""
import torch

from . import ops

_ALIGN = 64  # elements; keeps every parameter 128-byte aligned in bf16 (16-byte vector loads need 8)


class FusedAdam:
    """`model` (optional): modules exposing `packed_param_groups()` -> [(params, bind)] get those parameters laid
    out back to back in the arenas and `bind(param_view, grad_view)` called with flat views over the group (the
    attention block's q|k|v projections become one [3D, D] weight without a per-step torch.cat, and its gradient is
    written packed).  Every parameter is marked as a gradient sink (functional._sink): backward kernels accumulate
    into the arena directly."""

    def __init__(self, params, lr=5e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.01, clip_norm=0.0, model=None, pack=True):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no parameters")
        dev, dtype = self.params[0].device, self.params[0].dtype
        for p in self.params:
            if p.device != dev or p.dtype != dtype:
                raise ValueError("all parameters must share device and dtype")
        self.lr, self.betas, self.eps, self.weight_decay, self.clip_norm = lr, betas, eps, weight_decay, clip_norm
        groups = []
        if not pack:      # plain registration-order arena (tests: re-homing a checkpoint between layouts)
            pass
        elif model is not None:
            for m in model.modules():
                if hasattr(m, "packed_param_groups"):
                    groups.extend(m.packed_param_groups())
        else:
            # built from a bare parameter list (the fairseq Trainer hands its optimizer `model.parameters()` only,
            # trainer.py:275-316): the attention blocks tag their q|k|v parameters with their owner
            owners = []
            for p in self.params:
                m = getattr(p, "_wl_pack_owner", None)
                if m is not None and not any(m is o for o in owners):
                    owners.append(m)
            for m in owners:
                groups.extend(m.packed_param_groups())
        member = {}
        for gi, (gp, _bind) in enumerate(groups):
            if all(any(q is p for p in self.params) for q in gp) and all(q.numel() % 8 == 0 for q in gp):
                for q in gp:
                    member[id(q)] = gi
        # arena order: registration order, except that a packed group is placed whole at its first member
        order, placed = [], set()
        for p in self.params:
            gi = member.get(id(p))
            if gi is None:
                order.append([p])
            elif gi not in placed:
                placed.add(gi)
                order.append(list(groups[gi][0]))
        off_of = {}
        off = 0
        self._group_span = {}
        for blk in order:
            start = off
            for p in blk:
                off_of[id(p)] = off
                off += p.numel()
            if len(blk) > 1:
                self._group_span[member[id(blk[0])]] = (start, off)
            off = (off + _ALIGN - 1) // _ALIGN * _ALIGN
        self.offsets = [off_of[id(p)] for p in self.params]
        self.numel = off
        # what was packed (tests / logs: a deep-copied model whose tags were lost would show 0 here)
        self.packed_groups = len(self._group_span)
        self.flat_param = torch.zeros(off, dtype=dtype, device=dev)
        self.flat_grad = torch.zeros(off, dtype=dtype, device=dev)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = self.flat_param[o:o + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                p.grad = self.flat_grad[o:o + p.numel()].view_as(p)
                p._wl_sink = True
        for gi, (lo, hi) in self._group_span.items():
            groups[gi][1](self.flat_param[lo:hi], self.flat_grad[lo:hi])
        self.lowp = dtype != torch.float32
        self.master = self.flat_param.float() if self.lowp else self.flat_param
        self.exp_avg = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(off, dtype=torch.float32, device=dev)
        self.gnorm_sq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.step_count = 0
        # deferred gradient factor (fp16_optimizer.py:182-184 `_multiply_factor`): multiply_grads() and the data-parallel
        # wrapper's 1/world only update this number; the update kernel applies it (and the clip coefficient) on the fly, so
        # the gradient arena is read exactly once per step.  Reset by zero_grad() and after every step.
        self.pending_mult = 1.0
        self._norm_fresh = False

    def zero_grad(self):
        from . import functional
        functional.reset_sink_uses()
        self.flat_grad.zero_()
        self.pending_mult = 1.0
        self._norm_fresh = False
        for p, o in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + o * self.flat_grad.element_size():
                p.grad = self.flat_grad[o:o + p.numel()].view_as(p)

    def multiply_grads(self, c):
        """deferred: folded into the next step's gradient multiplier (and into grad_norm())"""
        self.pending_mult *= float(c)

    def compute_grad_norm_sq(self):
        """sum of squares of the (unscaled) gradient arena into self.gnorm_sq, on the device, no synchronisation"""
        from . import functional
        functional.flush_wgrad_groups()  # normally empty: groups fire during backward / at its end
        ops.sumsq(self.flat_grad, 1.0, out=self.gnorm_sq)
        self._norm_fresh = True
        return self.gnorm_sq

    def step(self, grad_mult=1.0, grad_mult_dev=None, max_norm=None):
        """grads are used as grad * grad_mult * pending_mult (* grad_mult_dev[0]); clipping (max_norm, default: the
        constructor's clip_norm) uses the norm of the scaled gradient"""
        if not self._norm_fresh:
            self.compute_grad_norm_sq()
        self.step_count += 1
        ops.adam_step(self.master, self.exp_avg, self.exp_avg_sq, self.flat_grad,
                      self.flat_param if self.lowp else None, lr=self.lr, beta1=self.betas[0], beta2=self.betas[1],
                      eps=self.eps, weight_decay=self.weight_decay, step=self.step_count,
                      grad_mult=grad_mult * self.pending_mult, grad_mult_dev=grad_mult_dev, gnorm_sq=self.gnorm_sq,
                      max_norm=self.clip_norm if max_norm is None else max_norm)
        self.pending_mult = 1.0
        self._norm_fresh = False
        # the update kernel writes the parameters through raw pointers: torch's in-place version counters do not move, so what
        # inference keeps derived from them (functional.eval_derived) is told here
        self._params_written()

    @staticmethod
    def _params_written():
        """every method that writes the parameters through the arena (views' version counters do not move) says so here"""
        from . import functional
        functional.invalidate_derived()

    def grad_norm(self, grad_mult=1.0):
        """host value of the (scaled) global gradient norm of the last step() / compute_grad_norm_sq() -- synchronises;
        logging only.  (After step() the deferred factor is already consumed: pass the factor used as grad_mult.)"""
        return float(self.gnorm_sq.sqrt().item()) * abs(grad_mult)

    def sync_master_from_params(self):
        """after the model's parameters were overwritten in place (load_state_dict into the arena views): the fp32 master
        copy follows, as FP16Optimizer rebuilds its fp32 parameters from the model's (fp16_optimizer.py:36-76)"""
        if self.lowp:
            self.master.copy_(self.flat_param)

    def _layout(self):
        """(offset, numel, shape) of every parameter in arena order of registration: what a checkpoint must agree on"""
        return [(int(o), int(p.numel()), tuple(p.shape)) for p, o in zip(self.params, self.offsets)]

    def state_dict(self, names=None):
        """Flat arenas + the layout they were written with.  `names` (optional, one per parameter, e.g. from
        model.named_parameters()) are stored too, so that a checkpoint can be re-homed into an arena laid out differently
        (another packing rule, a head added or removed) parameter by parameter."""
        sd = {"step": self.step_count, "master": self.master, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
              "lr": self.lr, "numel": self.numel, "layout": self._layout()}
        if names is not None:
            names = list(names)
            if len(names) != len(self.params):
                raise ValueError("names must have one entry per optimized parameter")
            sd["names"] = names
        return sd

    def load_state_dict(self, sd, names=None):
        """Same layout: three arena copies.  Different layout: matched by parameter name (both sides must carry names) and
        copied slice by slice; a parameter whose shape changed, or a layout mismatch without names, raises instead of
        silently misaligning the moments."""
        self.step_count = sd["step"]
        saved = sd.get("layout")
        same = saved is None or [tuple(x[:2]) + (tuple(x[2]),) for x in saved] == self._layout()
        if same and sd["master"].numel() == self.numel:
            self.master.copy_(sd["master"])
            self.exp_avg.copy_(sd["exp_avg"])
            self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        else:
            src_names = sd.get("names")
            if src_names is None or names is None:
                raise ValueError("optimizer state was saved with a different arena layout; pass parameter names on both "
                                 "sides (state_dict(names=...), load_state_dict(sd, names=...)) to re-home it")
            where = {n: (o, k, shp) for n, (o, k, shp) in zip(src_names, sd["layout"])}
            for n, p, o in zip(names, self.params, self.offsets):
                if n not in where:
                    continue  # new parameter: keeps its fresh state
                so, sk, sshape = where[n]
                if tuple(sshape) != tuple(p.shape):
                    raise ValueError("optimizer state of %s has shape %s, the parameter has %s" % (n, sshape, tuple(p.shape)))
                for dst, key in ((self.master, "master"), (self.exp_avg, "exp_avg"), (self.exp_avg_sq, "exp_avg_sq")):
                    dst[o:o + sk].copy_(sd[key][so:so + sk])
        if self.lowp:
            self.flat_param.copy_(self.master)
        self._params_written()   # (fp32 arenas too: master IS flat_param there and was just overwritten)

    def fairseq_state_dict(self):
        """Per-parameter state in the shape torch.optim / fairseq's Adam checkpoint it (optim/adam.py:176-195:
        state[i] = {step, exp_avg, exp_avg_sq}, param_groups with lr / betas / eps / weight_decay), fp32 views of the
        arenas -- what `optimizer_history` / `last_optimizer_state` of a reference checkpoint hold (trainer.py:373-411)."""
        state = {}
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            n = p.numel()
            state[i] = {"step": self.step_count, "exp_avg": self.exp_avg[o:o + n].view(p.shape),
                        "exp_avg_sq": self.exp_avg_sq[o:o + n].view(p.shape)}
        return {"state": state, "param_groups": [{"lr": self.lr, "betas": self.betas, "eps": self.eps,
                                                  "weight_decay": self.weight_decay, "amsgrad": False,
                                                  "params": list(range(len(self.params)))}]}

    def load_fairseq_state_dict(self, fsd):
        """inverse of fairseq_state_dict(): Adam moments of a reference checkpoint into the arenas (master weights come
        from the model's own state dict)"""
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            st = fsd["state"].get(i)
            if st is None:
                continue
            n = p.numel()
            self.exp_avg[o:o + n].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            self.step_count = int(st["step"])
        g = fsd["param_groups"][0]
        self.lr, self.betas, self.eps, self.weight_decay = g["lr"], tuple(g["betas"]), g["eps"], g["weight_decay"]


class DynamicLossScaler:
    """The reference's loss-scale schedule (optim/dynamic_loss_scaler.py:7-70), kept for `--fp16` recipes run with the
    fp16-as-bf16 switch (precision.py): multiply the loss by `loss_scale`; a non-finite gradient norm is an overflow --
    the scale is divided by `scale_factor` (not below `threshold`) once the share of overflowing updates since the last
    rescale reaches `tolerance`, and OverflowError tells the Trainer to skip the update (FloatingPointError once the scale
    would fall to `min_loss_scale`); `scale_window` updates after the last overflow the scale is multiplied again."""

    def __init__(self, init_scale=2.0 ** 15, scale_factor=2.0, scale_window=2000, tolerance=0.0, threshold=None,
                 min_loss_scale=1e-4):
        self.loss_scale, self.scale_factor, self.scale_window = init_scale, scale_factor, scale_window
        self.tolerance, self.threshold, self.min_loss_scale = tolerance, threshold, min_loss_scale
        self._iter, self._last_overflow_iter, self._last_rescale_iter, self._overflows_since_rescale = 0, -1, -1, 0

    def scale(self, outputs):
        return self.loss_scale * outputs

    def update(self):
        """after a completed update: grow the scale every `scale_window` updates without an overflow"""
        if (self._iter - self._last_overflow_iter) % self.scale_window == 0:
            self.loss_scale *= self.scale_factor
            self._last_rescale_iter = self._iter
        self._iter += 1

    def check_overflow(self, grad_norm):
        if not (grad_norm == float("inf") or grad_norm != grad_norm):
            return
        before = self.loss_scale
        since = self._iter - self._last_rescale_iter
        self._last_overflow_iter = self._iter
        self._overflows_since_rescale += 1
        if self._overflows_since_rescale / float(since) >= self.tolerance:
            self.loss_scale /= self.scale_factor
            if self.threshold is not None:
                self.loss_scale = max(self.loss_scale, self.threshold)
            self._last_rescale_iter = self._iter
            self._overflows_since_rescale = 0
        if self.loss_scale <= self.min_loss_scale:
            self.loss_scale = before
            raise FloatingPointError("Minimum loss scale reached ({}). Your loss is probably exploding. Try lowering the "
                                     "learning rate, using gradient clipping or increasing the batch size."
                                     .format(self.min_loss_scale))
        self._iter += 1
        raise OverflowError("setting loss scale to: " + str(self.loss_scale))


class FairseqFusedAdam:
    """The fairseq Trainer's optimizer seam over FusedAdam: every method `trainer.py` calls on `self.optimizer`
    (optim/fairseq_optimizer.py:97-130 and the bf16 wrapper optim/fp16_optimizer.py:106-289 that trainer.py:296-316
    instantiates) -- backward, all_reduce_grads, multiply_grads (deferred factor, fp16_optimizer.py:182-184),
    clip_grad_norm (device-side norm, returned as a device scalar; the clip coefficient is applied inside the update
    kernel), step, zero_grad, set_lr / get_lr, state_dict / load_state_dict in the layout fairseq's Adam checkpoints
    (optim/adam.py:176-195).  Plain Python on purpose: `fairseq_plugin` mixes it with `FairseqOptimizer` and registers it
    (`adam_mi355x`; `register(override=True)` also substitutes it for `optim.FP16Optimizer`) when fairseq is importable;
    without fairseq the same class drives the same call sequence (tests/test_trainer_seam_gpu.py).

    cfg: the optimizer config (`cfg.optimizer` of the reference: lr list, adam_betas, adam_eps, weight_decay)."""

    def __init__(self, cfg, params, scaler=None):
        self.cfg = cfg
        betas = getattr(cfg, "adam_betas", (0.9, 0.999))
        if isinstance(betas, str):
            import ast
            betas = ast.literal_eval(betas)
        lr = getattr(cfg, "lr", [1e-3])
        lr = float(lr[0]) if isinstance(lr, (list, tuple)) else float(lr)
        self.fused = FusedAdam(list(params), lr=lr, betas=tuple(float(b) for b in betas),
                               eps=float(getattr(cfg, "adam_eps", 1e-8)), weight_decay=float(getattr(cfg, "weight_decay", 0.0)),
                               clip_norm=0.0)
        self._mult_dev = None    # a factor that arrived as a device tensor (sample_size kept on the device)
        self._max_norm = 0.0
        # bf16: no loss scaling (fp16_optimizer.py:248-250); trainer.py:711,947 probe for it.  A DynamicLossScaler only when
        # an --fp16 recipe runs under the fp16-as-bf16 switch (precision.py)
        self.scaler = scaler
        from . import dp
        dp.bind_live_wrappers(self.fused)

    @classmethod
    def build_optimizer(cls, cfg, params, **kwargs):
        """the signature `optim.FP16Optimizer.build_optimizer(self.cfg, params)` is called with (trainer.py:312): the FULL
        config; the registry path `optim.build_optimizer(cfg.optimizer, params)` constructs the class directly"""
        common = getattr(cfg, "common", None)
        if getattr(common, "fp16", False) and not getattr(common, "bf16", False):
            from . import precision
            if not precision.fp16_as_bf16():
                raise NotImplementedError(precision.MESSAGE)
            # fp16_optimizer.py:241-268: the scaler of an --fp16 run, window from the data-parallel size when not given
            window = getattr(common, "fp16_scale_window", None)
            if window is None:
                dt, opt_ = getattr(cfg, "distributed_training", None), getattr(cfg, "optimization", None)
                uf = list(getattr(opt_, "update_freq", [1]))
                if len(uf) > 1:
                    raise ValueError("--fp16-scale-window must be given explicitly when using a custom --update-freq schedule")
                dp_size = int(getattr(dt, "distributed_world_size", 1) / getattr(common, "model_parallel_size", 1))
                window = int(2 ** 14 / dp_size / uf[0])
            scaler = DynamicLossScaler(init_scale=getattr(common, "fp16_init_scale", 2 ** 7), scale_window=window,
                                       tolerance=getattr(common, "fp16_scale_tolerance", 0.0),
                                       threshold=getattr(common, "threshold_loss_scale", None),
                                       min_loss_scale=getattr(common, "min_loss_scale", 1e-4))
            return cls(cfg.optimizer, params, scaler=scaler)
        return cls(cfg.optimizer, params)

    # -- what the Trainer reads ------------------------------------------------------------------------------------
    @property
    def optimizer(self):
        return self.fused

    @property
    def optimizer_config(self):
        return {"lr": self.fused.lr, "betas": self.fused.betas, "eps": self.fused.eps, "weight_decay": self.fused.weight_decay}

    @property
    def params(self):
        return iter(self.fused.params)

    @property
    def param_groups(self):
        f = self.fused
        return [{"lr": f.lr, "betas": f.betas, "eps": f.eps, "weight_decay": f.weight_decay, "amsgrad": False,
                 "params": f.params}]

    supports_memory_efficient_fp16 = False
    supports_step_with_scale = True
    supports_groups = False
    supports_flat_params = True

    def get_lr(self):
        return self.fused.lr

    def set_lr(self, lr):
        self.fused.lr = float(lr)

    # -- the step sequence of trainer.py:697-860 -------------------------------------------------------------------
    def backward(self, loss):
        if self.scaler is not None:
            loss = self.scaler.scale(loss)   # fp16_optimizer.py:112-120; 1 / loss_scale is in the deferred factor (zero_grad)
        loss.backward()

    def all_reduce_grads(self, module):
        # bound late (the Trainer wraps the model before it builds the optimizer) or bound to an arena this optimizer has
        # replaced (the Trainer rebuilt its optimizer: trainer.py load_checkpoint / reinitialize)
        if hasattr(module, "bind_optimizer") and getattr(module, "_optimizer", self.fused) is not self.fused:
            module.bind_optimizer(self.fused)
        if hasattr(module, "all_reduce_grads"):
            module.all_reduce_grads()

    def multiply_grads(self, c):
        if torch.is_tensor(c):
            c = c.detach().to(device=self.fused.flat_grad.device, dtype=torch.float32).reshape(1)
            self._mult_dev = c if self._mult_dev is None else self._mult_dev * c
        else:
            self.fused.multiply_grads(c)

    def clip_grad_norm(self, max_norm, aggregate_norm_fn=None):
        """norm of the gradient as the update will see it (deferred factors included), as a 0-dim device tensor -- no host
        synchronisation here; the Trainer's own isfinite check (trainer.py:817) is the first reader.  Clipping itself
        happens inside the update kernel with the same coefficient formula (max_norm / (norm + 1e-6), clamped to 1)."""
        if aggregate_norm_fn is not None:
            raise NotImplementedError("aggregate_norm_fn (fully_sharded) is not part of this path")
        n = self.fused.compute_grad_norm_sq().sqrt() * abs(self.fused.pending_mult)
        if self._mult_dev is not None:
            n = n * self._mult_dev.abs()
        self._max_norm = float(max_norm) if max_norm else 0.0
        if self.scaler is not None:
            # fp16_optimizer.py:199-206: with a loss scaler the overflow check reads the norm on the host (the reference
            # synchronises here too) and raises OverflowError before anything is updated
            self.scaler.check_overflow(float(n))
        return n.reshape(())

    def step(self, closure=None, scale=1.0, groups=None):
        if closure is not None or groups is not None:
            raise NotImplementedError("closure / parameter groups are not supported by the fused update")
        self.fused.step(grad_mult=1.0 / float(scale), grad_mult_dev=self._mult_dev, max_norm=self._max_norm)
        self._mult_dev, self._max_norm = None, 0.0
        if self.scaler is not None:
            self.scaler.update()             # fp16_optimizer.py:218-219

    def zero_grad(self):
        self.fused.zero_grad()
        self._mult_dev, self._max_norm = None, 0.0
        if self.scaler is not None:          # fp16_optimizer.py:238-239: the next backward's gradients carry loss_scale
            self.fused.pending_mult = 1.0 / float(self.scaler.loss_scale)

    # -- checkpoints (trainer.py:373-411, 511-543) ------------------------------------------------------------------
    def state_dict(self):
        sd = self.fused.fairseq_state_dict()
        if self.scaler is not None:          # fp16_optimizer.py:79: a resumed --fp16 run continues at its loss scale
            sd["loss_scale"] = self.scaler.loss_scale
        return sd

    def load_state_dict(self, state_dict, optimizer_overrides=None):
        if "loss_scale" in state_dict:       # fp16_optimizer.py:90-91 (ignored without a scaler, as a --bf16 resume does)
            state_dict = dict(state_dict)
            ls = state_dict.pop("loss_scale")
            if self.scaler is not None:
                self.scaler.loss_scale = ls
        self.fused.load_fairseq_state_dict(state_dict)
        self.fused.sync_master_from_params()
        if optimizer_overrides:
            for k, v in optimizer_overrides.items():
                if k == "lr":
                    self.fused.lr = float(v)
                elif k == "betas":
                    self.fused.betas = tuple(v)
                elif k == "eps":
                    self.fused.eps = float(v)
                elif k == "weight_decay":
                    self.fused.weight_decay = float(v)

    def broadcast_global_state_dict(self, state_dict):
        return state_dict
