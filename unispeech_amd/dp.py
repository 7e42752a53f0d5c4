"""Data-parallel gradient reduction over RCCL / xGMI.

The reference recipes use LegacyDistributedDataParallel: copy every gradient into one flat buffer *after* the
whole backward, divide by world size, one blocking all-reduce (src/fairseq/distributed/
legacy_distributed_data_parallel.py:76-165).  Here gradients already live in one flat arena (optim.FusedAdam), so
there is nothing to pack: the arena is cut into buckets in reverse parameter order, a post-accumulate hook counts
the gradients of a bucket as backward produces them, and a finished bucket is all-reduced immediately on a side
HIP stream while the rest of backward keeps the compute stream busy.  xGMI is point-to-point (7 links x ~153 GB/s
per GPU): a ring all-reduce is bound by one link, so buckets are kept large (default 32 MiB -> ~6 collectives for
WavLM-Base) -- the goal is overlap, not message count.

Semantics kept from the reference wrapper: `no_sync()` (gradient accumulation), `all_reduce_grads()` as the
explicit completion point the fairseq Trainer calls (trainer.py:781-785), and the AVERAGE over ranks as its result
(legacy_distributed_data_parallel.py:107-108 divides by the world size before the all-reduce; the Trainer then
multiplies by world / sample_size, trainer.py:796-801).  The division costs no pass over the arena: the collectives SUM,
and `DataParallelWavLM.all_reduce_grads()` folds 1/world into the optimizer's deferred gradient factor
(`FusedAdam.pending_mult`, the counterpart of fp16_optimizer.py's `_multiply_factor`), which the update kernel and the
gradient norm apply on the fly -- as the reference's own mixed-precision wrapper does with multiply_grads ("inspecting
model.parameters() ... may still show the original, unscaled gradients", trainer.py:793-795).
"""
import weakref
from contextlib import contextmanager

import torch
import torch.distributed as dist


def reserved_channels():
    """CUs the persistent GEMM grids leave to RCCL == the number of channels RCCL should be limited to (WAVLM_DP_RESERVED_CUS)"""
    import os
    return max(0, min(64, int(os.environ.get("WAVLM_DP_RESERVED_CUS", "6"))))


def cap_rccl_channels(world=None, log=False):
    """NCCL_MAX_NCHANNELS = the CUs the persistent GEMM grids leave free, set BEFORE the communicator exists.  The variable is
    process-wide (it caps every RCCL communicator of the process, not only the gradient all-reduce), so: only for a
    data-parallel run (`world` > 1, or WORLD_SIZE > 1 in the launcher's environment when `world` is None), never over a
    value the user exported, and said out loud -- including when it is too late (torch.distributed already initialised:
    RCCL may then take more CUs than the persistent grids leave free).  Returns the value in force or None.
    The cap is a HYPOTHESIS (one channel = one workgroup = one CU; 332 MB per rank and step through 6 channels is an
    untested bandwidth assumption: DESIGN.md section 5); `bench.py --gpus N` measures both settings when the wait is long."""
    import os
    import sys
    n = reserved_channels()
    if world is None:
        world = int(os.environ.get("WORLD_SIZE", "1") or 1)
    if n <= 0 or world <= 1:
        return os.environ.get("NCCL_MAX_NCHANNELS")
    if "NCCL_MAX_NCHANNELS" in os.environ:
        return os.environ["NCCL_MAX_NCHANNELS"]
    if dist.is_available() and dist.is_initialized():
        if log:
            print("[unispeech_amd] torch.distributed is already initialised: NCCL_MAX_NCHANNELS=%d can no longer be applied "
                  "(RCCL may occupy more CUs than WAVLM_DP_RESERVED_CUS=%d leaves free)" % (n, n), file=sys.stderr)
        return None
    os.environ["NCCL_MAX_NCHANNELS"] = str(n)
    if log:
        print("[unispeech_amd] NCCL_MAX_NCHANNELS=%d for this process (= WAVLM_DP_RESERVED_CUS; export NCCL_MAX_NCHANNELS "
              "yourself to override)" % n, file=sys.stderr)
    return str(n)


class NativeTransport:
    """The C-ABI RCCL reducer (csrc/dp_rccl.hip: wavlm_dp_init / _bucket_ready / _finish) as the transport of GradReducer --
    what a non-Python host would call, driven from here so that it is exercised: WAVLM_DP_NATIVE=1.  torch.distributed only
    carries the 128-byte ncclUniqueId from rank 0 to the other ranks (the bootstrap any host has to supply).  One communicator
    per process; buckets are all-reduced (SUM) in place on the library's own high-priority stream, each behind an event on the
    compute stream; `finish` makes the compute stream wait for them.  No torch communication stream, no Work objects."""

    def __init__(self, process_group, device):
        import ctypes
        from . import _lib
        self.L = _lib.lib()
        rank, world = dist.get_rank(process_group), dist.get_world_size(process_group)
        buf = ctypes.create_string_buffer(128)
        if rank == 0:
            rc = self.L.wavlm_dp_unique_id(buf)
            if rc != 0:
                raise RuntimeError("wavlm_dp_unique_id failed (%d): RCCL could not be loaded" % rc)
        box = [bytes(buf.raw) if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(process_group, 0) if process_group is not None else 0,
                                   group=process_group)
        self._id = ctypes.create_string_buffer(box[0], 128)
        with torch.cuda.device(device):
            rc = self.L.wavlm_dp_init(rank, world, self._id, 0)
        if rc != 0:
            raise RuntimeError("wavlm_dp_init failed (%d)" % rc)

    def bucket_ready(self, view):
        from . import ops
        rc = self.L.wavlm_dp_bucket_ready(ops.ptr(view), view.numel(), ops.dt(view), ops.stream())
        if rc != 0:
            raise RuntimeError("wavlm_dp_bucket_ready failed (%d)" % rc)

    def finish(self):
        from . import ops
        rc = self.L.wavlm_dp_finish(ops.stream())
        if rc != 0:
            raise RuntimeError("wavlm_dp_finish failed (%d)" % rc)

    def close(self):
        self.L.wavlm_dp_destroy()


def native_transport_requested():
    import os
    return os.environ.get("WAVLM_DP_NATIVE", "0") == "1"


class GradReducer:
    def __init__(self, params, flat_grad, offsets, process_group=None, bucket_bytes=32 << 20, transport=None):
        self.params = list(params)
        self.flat_grad = flat_grad
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # WAVLM_DP_FORCE=1 (bench.py --force-dp): a one-rank group still runs the whole machinery -- buckets, side stream,
        # listeners, reserved CUs, the collective itself (an in-place copy for RCCL at world 1) --, which prices what the
        # data-parallel path costs a rank before any byte crosses a link
        import os
        self.enabled = self.world > 1 or (dist.is_initialized() and os.environ.get("WAVLM_DP_FORCE", "0") == "1")
        self.sync = True
        self.is_cuda = flat_grad.is_cuda
        # transport: None = torch.distributed collectives on a side stream of torch's; an object with bucket_ready(view) /
        # finish() / close() = the C-ABI reducer (NativeTransport; WAVLM_DP_NATIVE=1) or a test double
        self.transport = transport
        if self.transport is None and self.enabled and self.is_cuda and native_transport_requested():
            self.transport = NativeTransport(process_group, flat_grad.device)
        self.comm_stream = (torch.cuda.Stream(device=flat_grad.device)
                            if (self.is_cuda and self.enabled and self.transport is None) else None)
        if self.is_cuda and self.enabled:
            # a persistent 256-block GEMM owns every CU (128 KiB of LDS + all VGPRs per block), so an all-reduce launched
            # on the side stream would only start at the next kernel boundary: persistent grids shrink to 256 - n and
            # leave n CUs to the RCCL kernels while backward is still running.  Default 6: the activation GEMMs at
            # 32 x 15 s / 32 x 20 s have 125 row tiles, i.e. 250 / 500 / 750 / 1000 / 2000 tiles -- whole rounds of a
            # 250-block grid; measured on one rank (profiles/r04/reserved_cus_*.txt).  The RCCL kernels must not
            # take MORE than n CUs either (a channel = a workgroup = a CU; the next persistent grid would then run a second
            # round): reserved_channels() is what the launcher puts into NCCL_MAX_NCHANNELS before the communicator exists.
            from . import ops
            ops.set_reserved_cus(reserved_channels())
        # buckets = disjoint contiguous arena ranges, cut walking the arena from its END (gradients of the last layers
        # are produced first).  The walk is in arena-offset order, not parameter order: the optimizer lays packed
        # groups (q|k|v) out of registration order, and ranges cut by parameter index could overlap there.
        esize = flat_grad.element_size()
        self.buckets = []  # dict(lo, hi, n)
        self.bucket_of = {}
        cur = None
        top = flat_grad.numel()
        for idx in sorted(range(len(self.params)), key=lambda i: -offsets[i]):
            o = offsets[idx]
            if cur is None:
                cur = {"lo": o, "hi": top, "n": 0}
            cur["lo"] = o
            cur["n"] += 1
            self.bucket_of[idx] = len(self.buckets)
            if (cur["hi"] - cur["lo"]) * esize >= bucket_bytes:
                self.buckets.append(cur)
                top = cur["lo"]
                cur = None
        if cur is not None:
            cur["lo"] = 0
            self.buckets.append(cur)
        elif self.buckets:
            self.buckets[-1]["lo"] = 0
        self._ready = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._seen = [False] * len(self.params)
        self._next = 0  # buckets are launched strictly in index order on every rank (see _mark)
        self._works = []
        # record_trace=True: every bucket launch leaves (bucket, timing event recorded on the COMPUTE stream at the point of
        # the launch) in `trace` -- tests / tools/dp_overlap.py compare it with an event recorded when backward returns
        self.record_trace = False
        self.trace = []
        # record_timing=True (bench.py --gpus N, diagnostic steps only): per bucket (index, bytes, launch event on the compute
        # stream, start / end events around the collective on the communication stream) in `timing` -- how early each bucket
        # went out, how long its all-reduce took, what the last one leaves exposed behind backward
        self.record_timing = False
        self.timing = []
        # arena element offsets (sorted) for sink notifications: backward kernels that accumulate straight into the
        # arena (functional._sink) report the slice they wrote; a packed q|k|v slice covers three parameters
        self._offsets = list(offsets)
        self._order = sorted(range(len(self.params)), key=lambda i: self._offsets[i])
        self._sorted_off = [self._offsets[i] for i in self._order]
        self._hooks = []
        self._listener = None
        if self.enabled:
            for idx, p in enumerate(self.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(idx)))
            from . import functional
            self._listener = self._sink_written   # (one bound-method object: `remove` below must find this very one)
            functional.SINK_LISTENERS.append(self._listener)

    def close(self):
        """detach from the parameters and from the kernels' sink notifications: called when the wrapper is re-bound to
        another optimizer arena (the Trainer rebuilds its optimizer in load_checkpoint / reinitialize) -- a reducer left
        behind would keep launching collectives on a dead arena"""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if self._listener is not None:
            from . import functional
            try:
                functional.SINK_LISTENERS.remove(self._listener)
            except ValueError:
                pass
            self._listener = None
        if self.transport is not None and hasattr(self.transport, "close"):
            self.transport.close()
            self.transport = None
        self.enabled = False

    def _mark(self, idx):
        """parameter idx has its gradient of this backward in the arena (autograd hook or sink notification; idempotent)"""
        if not (self.enabled and self.sync) or self._seen[idx]:
            return
        self._seen[idx] = True
        self._ready[self.bucket_of[idx]] += 1
        # Collectives must be issued in the SAME order on every rank, but readiness is data dependent: with
        # encoder_layerdrop > 0 each rank drops different layers (the numpy streams diverge with the batch-dependent mask
        # draws), a dropped layer's bucket never becomes ready during backward, and a rank that skipped it would pair its
        # next all-reduce with a different bucket on its peers (hang, or silently mixed gradients).  Hence strict index
        # order, as torch DDP does: bucket b starts only when it is complete AND all buckets before it have started;
        # whatever is held back goes out in finish(), still in index order.  Bucket 0 is the END of the arena (the last
        # layers), i.e. the order backward produces gradients in, so nothing is delayed when no layer is dropped.
        while self._next < len(self.buckets) and self._ready[self._next] == self.buckets[self._next]["n"]:
            self._launch(self._next)
            self._next += 1

    def _make_hook(self, idx):
        from . import functional

        def hook(p):
            if functional.grad_write_deferred(p.grad):
                return  # its weight gradient is queued for a grouped launch; the sink notification will mark it
            self._mark(idx)
        return hook

    def _sink_written(self, t):
        if not (self.enabled and self.sync):
            return
        base, esize = self.flat_grad.data_ptr(), self.flat_grad.element_size()
        lo = (t.data_ptr() - base) // esize
        if lo < 0 or lo >= self.flat_grad.numel():
            return
        hi = lo + t.numel()
        import bisect
        k = bisect.bisect_left(self._sorted_off, lo)
        while k < len(self._sorted_off) and self._sorted_off[k] < hi:
            self._mark(self._order[k])
            k += 1

    def _launch(self, b):
        bk = self.buckets[b]
        view = self.flat_grad[bk["lo"]:bk["hi"]]
        self._launched[b] = True
        if self.transport is not None:
            if self.record_trace:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(torch.cuda.current_stream())
                self.trace.append((b, ev))
            self.transport.bucket_ready(view)
        elif self.comm_stream is not None:
            ev = torch.cuda.Event(enable_timing=self.record_trace or self.record_timing)
            ev.record(torch.cuda.current_stream())
            if self.record_trace:
                self.trace.append((b, ev))
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                if self.record_timing:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(self.comm_stream)
                    w = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                    w.wait()     # the communication stream waits for the collective's own stream: e1 then marks its end
                    e1.record(self.comm_stream)
                    self.timing.append((b, view.numel() * view.element_size(), ev, e0, e1))
                    self._works.append(w)
                else:
                    self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        else:
            self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    @contextmanager
    def no_sync(self):
        old = self.sync
        self.sync = False
        try:
            yield
        finally:
            self.sync = old

    def finish(self):
        """Complete the reduction of this step: launch buckets whose hooks never all fired (unused parameters,
        accumulation steps), then make the compute stream wait for every collective.  The arena holds the SUM over
        ranks afterwards; `self.scale` (1/world) turns it into the average (DataParallelWavLM folds it into the
        optimizer's deferred factor)."""
        if not self.enabled:
            return
        from . import functional
        functional.flush_wgrad_groups()
        for b in range(self._next, len(self.buckets)):  # index order; everything before _next is already in flight
            assert not self._launched[b]
            self._launch(b)
        for w in self._works:
            w.wait()
        if self.transport is not None:
            self.transport.finish()
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        self._works = []
        self._ready = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._seen = [False] * len(self.params)
        self._next = 0

    @property
    def scale(self):
        return 1.0 / self.world


_LIVE_WRAPPERS = weakref.WeakSet()


def bind_live_wrappers(optimizer):
    """called by an optimizer front-end built AFTER the model was wrapped (the fairseq Trainer wraps the model in its
    `model` property, trainer.py:250-261, and builds the optimizer lazily from `self.model.parameters()` afterwards):
    every unbound wrapper whose module owns the optimizer's parameters gets its reducer now"""
    mine = {id(p) for p in optimizer.params}
    for w in list(_LIVE_WRAPPERS):
        # also a wrapper bound to an OLDER arena of the same parameters: the Trainer builds its optimizer again after
        # load_checkpoint / reinitialize, p.grad then points into the new arena and the old reducer would go on reducing
        # the old one (ranks diverge silently)
        if w._optimizer is not optimizer and any(id(p) in mine for p in w.module.parameters()):
            w.bind_optimizer(optimizer)


class DataParallelWavLM(torch.nn.Module):
    """LegacyDDP-shaped wrapper (forward / no_sync / all_reduce_grads / attribute pass-through, state_dict of the wrapped
    module as fairseq's ModuleProxyWrapper gives it, distributed/module_proxy_wrapper.py:42-48) around a model whose
    gradients live in a FusedAdam arena.  `optimizer` may be given later (`bind_optimizer`): the Trainer wraps the model
    before it builds the optimizer."""

    def __init__(self, module, optimizer=None, process_group=None, bucket_bytes=32 << 20):
        super().__init__()
        self.module = module
        self.reducer = None
        self._optimizer = None
        self._pg, self._bucket_bytes = process_group, bucket_bytes
        self._accumulate = False
        _LIVE_WRAPPERS.add(self)
        if optimizer is not None:
            self.bind_optimizer(optimizer)

    def bind_optimizer(self, optimizer):
        """optimizer: a FusedAdam (or a front-end exposing it as `.fused`)"""
        optimizer = getattr(optimizer, "fused", optimizer)
        if self.reducer is not None:
            if self._optimizer is optimizer:
                return
            # a new arena for the same model (optimizer rebuilt): the old reducer's hooks and listener go, a new one is cut
            self.reducer.close()
            self.reducer = None
        self._optimizer = optimizer
        self.reducer = GradReducer(optimizer.params, optimizer.flat_grad, optimizer.offsets, self._pg, self._bucket_bytes)

    def forward(self, *a, **k):
        return self.module(*a, **k)

    @contextmanager
    def no_sync(self):
        if self.reducer is None:  # nothing to hold back yet: the first all_reduce_grads() reduces the whole arena
            yield
            return
        with self.reducer.no_sync():
            yield

    def all_reduce_grads(self):
        """gradients = AVERAGE over ranks afterwards, as seen by the optimizer (module docstring): the arena holds the
        sum, 1/world rides in the optimizer's deferred factor"""
        if self.reducer is None:
            raise RuntimeError(
                "DataParallelWavLM.all_reduce_grads(): no optimizer arena bound.  Gradients of this path live in a "
                "FusedAdam arena; build the optimizer through unispeech_amd (optimizer `adam_mi355x`, or "
                "fairseq_plugin.register(override=True) with --bf16) or call bind_optimizer(FusedAdam) first")
        self.reducer.finish()
        if self.reducer.enabled:
            self._optimizer.multiply_grads(self.reducer.scale)

    def state_dict(self, *args, **kwargs):
        return self.module.state_dict(*args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        return self.module.load_state_dict(*args, **kwargs)

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(super().__getattr__("module"), name)


def distributed_model(args, model, process_group, device, fallback=None):
    """Stand-in for `DistributedFairseqModel(args, model, process_group, device)` (models/distributed_fairseq_model.py:
    32-137), installed by `fairseq_plugin.register(override=True)`: models of this package get `DataParallelWavLM` for the
    ddp backends whose contract is "explicit or implicit all-reduce of averaged gradients" (legacy_ddp / no_c10d -- what the
    WavLM / HuBERT recipes run -- and c10d / pytorch_ddp); everything else goes to the reference's own function."""
    from .pretrain import WavLMPretrainModel
    ours = isinstance(model, WavLMPretrainModel)
    if not ours:
        try:
            from .wav2vec2 import Wav2Vec2Model
            ours = isinstance(model, Wav2Vec2Model)
        except ImportError:
            pass
    backend = getattr(args, "ddp_backend", "legacy_ddp")
    if ours and backend in {"legacy_ddp", "no_c10d", "c10d", "pytorch_ddp"}:
        return DataParallelWavLM(model.to(device), None, process_group)
    if fallback is None:
        raise ValueError("ddp_backend %r is not served by unispeech_amd.dp and no fallback was given" % backend)
    return fallback(args, model, process_group, device)
