"""world_size-2 test of the bucketed gradient reducer on the gloo backend (CPU): gradients equal the sum over ranks,
no_sync() accumulates locally, finish() also reduces buckets whose hooks never fired (unused parameters), and the
reducer's scale turns the sum into LegacyDDP's average."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(seed):
    torch.manual_seed(seed)
    net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.Tanh(), torch.nn.Linear(32, 8))
    unused = torch.nn.Parameter(torch.randn(40))
    params = list(net.parameters()) + [unused]
    # arena order differs from registration order (the optimizer packs q|k|v groups out of order)
    offsets, off = [0] * len(params), 0
    for i in (2, 0, 3, 1, 4):
        offsets[i] = off
        off += (params[i].numel() + 7) // 8 * 8
    flat = torch.zeros(off)
    for p, o in zip(params, offsets):
        p.grad = flat[o:o + p.numel()].view_as(p)
    return net, params, offsets, flat


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unispeech_amd.dp import GradReducer
    net, params, offsets, flat = _build(0)  # same weights on both ranks
    red = GradReducer(params, flat, offsets, bucket_bytes=1024)  # several buckets
    assert len(red.buckets) >= 2
    spans = sorted((b["lo"], b["hi"]) for b in red.buckets)  # disjoint, covering the arena
    assert spans[0][0] == 0 and spans[-1][1] == flat.numel()
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    x = torch.randn(4, 16, generator=torch.Generator().manual_seed(100 + rank))
    # local reference gradients of both ranks, computed independently
    refs = []
    for r in range(world):
        n2, p2, o2, f2 = _build(0)
        xr = torch.randn(4, 16, generator=torch.Generator().manual_seed(100 + r))
        n2(xr).pow(2).sum().backward()
        refs.append(f2.clone())
    # 1) plain step
    net(x).pow(2).sum().backward()
    red.finish()
    ok1 = torch.allclose(flat, refs[0] + refs[1], atol=1e-6)
    ok_avg = torch.allclose(flat * red.scale, (refs[0] + refs[1]) / world, atol=1e-6)
    # 2) accumulation: first micro-batch under no_sync, second synced -> sum over ranks of 2x local grad
    flat.zero_()
    with red.no_sync():
        net(x).pow(2).sum().backward()
    local_only = torch.allclose(flat, refs[rank], atol=1e-6)
    net(x).pow(2).sum().backward()
    red.finish()
    ok2 = torch.allclose(flat, 2 * (refs[0] + refs[1]), atol=1e-5)
    # 3) sink notifications (backward kernels that accumulate straight into the arena): a slice used twice in one
    #    forward only counts as complete after its second accumulation; a slice spanning two parameters marks both
    from unispeech_amd import functional as Fn
    flat.zero_()
    Fn.reset_sink_uses()
    w0 = flat[offsets[0]:offsets[0] + params[0].numel()]
    span = flat[offsets[3]:offsets[1] + params[1].numel()]  # parameters 3 and 1 are adjacent in the arena: one packed slice
    Fn._sink_use(None, w0)
    Fn._sink_use(None, w0)
    Fn._sink_use(None, span)
    Fn._sink_written(w0)
    first = red._seen[0]
    Fn._sink_written(w0)
    second = red._seen[0]
    Fn._sink_written(span)
    both = red._seen[3] and red._seen[1]
    ok3 = (not first) and second and both
    # a parameter whose weight gradient is queued for a grouped launch: autograd's hook fires early and must not count
    hook4 = red._make_hook(4)
    Fn.WgradGroup.deferred.add(params[4].grad.data_ptr())
    hook4(params[4])
    early = red._seen[4]
    Fn.WgradGroup.deferred.discard(params[4].grad.data_ptr())
    hook4(params[4])
    ok3 = ok3 and (not early) and red._seen[4]
    red.finish()
    ok3 = ok3 and not any(red._seen)
    # 4) rank-dependent readiness (encoder_layerdrop: each rank skips different layers, so different buckets stay
    #    un-ready during backward): the order in which all-reduces are ISSUED must still be identical on all ranks
    flat.zero_()
    issued = []
    orig_launch = red._launch
    red._launch = lambda b: (issued.append(b), orig_launch(b))[1]
    order = sorted(range(len(params)), key=lambda i: -offsets[i])      # the order backward produces gradients in
    skip = order[1] if rank == 0 else order[-2]                          # each rank "drops" a different parameter
    for i in order:
        if i != skip:
            params[i].grad.add_(float(rank + 1))
            red._mark(i)
    mid = list(issued)
    red.finish()
    ok4 = issued == list(range(len(red.buckets))) and mid == list(range(len(mid)))
    want = torch.full_like(flat, 3.0)
    for r_, sk in ((0, order[1]), (1, order[-2])):
        want[offsets[sk]:offsets[sk] + params[sk].numel()] -= float(r_ + 1)
    pad = torch.ones_like(flat, dtype=torch.bool)
    for p_, o_ in zip(params, offsets):
        pad[o_:o_ + p_.numel()] = False
    ok4 = ok4 and torch.allclose(flat[~pad], want[~pad])
    red._launch = orig_launch
    q.put((rank, ok1, ok_avg, local_only, ok2, ok3, ok4))
    dist.destroy_process_group()


def test_grad_reducer_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res:
        assert all(r[1:]), r


def _seam_worker(rank, world, port, q):
    """the Trainer's order of operations on the wrapper (CPU arenas: everything but the update kernel): wrap the model
    FIRST (trainer.py:250-261), build the optimizer front-end from the bare parameter list afterwards (trainer.py:275-316),
    backward under the wrapper, optimizer.all_reduce_grads(model) (trainer.py:781-785), multiply_grads(world / sample_size)
    (trainer.py:796-801)"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from types import SimpleNamespace
        from unispeech_amd.dp import DataParallelWavLM
        from unispeech_amd.optim import FairseqFusedAdam
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.Tanh(), torch.nn.Linear(32, 8))
        ddp = DataParallelWavLM(net, None, bucket_bytes=1024)
        assert ddp.reducer is None
        cfg = SimpleNamespace(lr=[1e-3], adam_betas="(0.9, 0.98)", adam_eps=1e-6, weight_decay=0.0)
        opt = FairseqFusedAdam(cfg, [p for p in ddp.parameters() if p.requires_grad])
        bound = ddp.reducer is not None and len(ddp.reducer.buckets) >= 2
        f = opt.fused
        x = torch.randn(4, 16, generator=torch.Generator().manual_seed(100 + rank))
        # local gradients of every rank, computed independently (same weights everywhere)
        refs = []
        for r in range(world):
            torch.manual_seed(0)
            n2 = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.Tanh(), torch.nn.Linear(32, 8))
            n2(torch.randn(4, 16, generator=torch.Generator().manual_seed(100 + r))).pow(2).sum().backward()
            refs.append([p.grad.clone() for p in n2.parameters()])
        opt.zero_grad()
        opt.backward(ddp(x).pow(2).sum())
        opt.all_reduce_grads(ddp)
        # the arena holds the SUM; the optimizer sees the AVERAGE (legacy_distributed_data_parallel.py:107-108)
        ok_sum = all(torch.allclose(p.grad, refs[0][i] + refs[1][i], atol=1e-6) for i, p in enumerate(net.parameters()))
        ok_avg = abs(f.pending_mult - 1.0 / world) < 1e-12
        sample_size = 40.0
        opt.multiply_grads(world / sample_size)
        ok_mult = abs(f.pending_mult - 1.0 / sample_size) < 1e-12      # == (sum of gradients) / sample_size at the update
        # accumulation (update_freq 2): first micro-batch under no_sync, all_reduce_grads once
        opt.zero_grad()
        ok_reset = f.pending_mult == 1.0
        with ddp.no_sync():
            opt.backward(ddp(x).pow(2).sum())
        local_only = all(torch.allclose(p.grad, refs[rank][i], atol=1e-6) for i, p in enumerate(net.parameters()))
        opt.backward(ddp(x).pow(2).sum())
        opt.all_reduce_grads(ddp)
        ok_acc = all(torch.allclose(p.grad, 2 * (refs[0][i] + refs[1][i]), atol=1e-5) for i, p in enumerate(net.parameters()))
        ok_acc = ok_acc and abs(f.pending_mult - 1.0 / world) < 1e-12
        keys = list(ddp.state_dict().keys()) == list(net.state_dict().keys())
        q.put((rank, bound, ok_sum, ok_avg, ok_mult, ok_reset, local_only, ok_acc, keys, None))
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_trainer_order_on_the_wrapper_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_seam_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[-1] is None and all(r[1:-1]), r


def _rebuild_worker(rank, world, port, q):
    """the Trainer builds its optimizer a SECOND time over the same wrapped model (trainer.py load_checkpoint: 'rebuild
    optimizer after loading model'; reinitialize): the wrapper must follow to the new arena -- the old reducer's hooks and sink
    listener go, the new arena is what gets reduced, and 1/world lands on the NEW optimizer's deferred factor"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from types import SimpleNamespace
        from unispeech_amd import functional as Fn
        from unispeech_amd.dp import DataParallelWavLM
        from unispeech_amd.optim import FairseqFusedAdam
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.Tanh(), torch.nn.Linear(32, 8))
        ddp = DataParallelWavLM(net, None, bucket_bytes=1024)
        cfg = SimpleNamespace(lr=[1e-3], adam_betas="(0.9, 0.98)", adam_eps=1e-6, weight_decay=0.0)
        params = [p for p in ddp.parameters() if p.requires_grad]
        opt1 = FairseqFusedAdam(cfg, params)
        red1 = ddp.reducer
        n_listeners_1 = len(Fn.SINK_LISTENERS)
        opt2 = FairseqFusedAdam(cfg, params)                       # rebuilt: a new arena, p.grad re-pointed
        red2 = ddp.reducer
        rebound = red2 is not red1 and ddp._optimizer is opt2.fused and red2.flat_grad is opt2.fused.flat_grad
        old_detached = (not red1.enabled) and not red1._hooks and len(Fn.SINK_LISTENERS) == n_listeners_1
        x = torch.randn(4, 16, generator=torch.Generator().manual_seed(100 + rank))
        refs = []
        for r in range(world):
            torch.manual_seed(0)
            n2 = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.Tanh(), torch.nn.Linear(32, 8))
            n2(torch.randn(4, 16, generator=torch.Generator().manual_seed(100 + r))).pow(2).sum().backward()
            refs.append([p.grad.clone() for p in n2.parameters()])
        opt2.zero_grad()
        opt2.backward(ddp(x).pow(2).sum())
        opt2.all_reduce_grads(ddp)
        in_new_arena = all(p.grad.data_ptr() >= opt2.fused.flat_grad.data_ptr() for p in net.parameters())
        ok_sum = all(torch.allclose(p.grad, refs[0][i] + refs[1][i], atol=1e-6) for i, p in enumerate(net.parameters()))
        ok_factor = abs(opt2.fused.pending_mult - 1.0 / world) < 1e-12 and opt1.fused.pending_mult == 1.0
        # a wrapper that was handed the stale optimizer explicitly and meets the new one only in all_reduce_grads()
        ddp.reducer.close()   # (one live wrapper per model: two reducers on one arena would reduce it twice)
        ddp3 = DataParallelWavLM(net, opt1.fused, bucket_bytes=1024)
        opt2.zero_grad()
        opt2.backward(ddp3(x).pow(2).sum())
        opt2.all_reduce_grads(ddp3)
        ok_late = ddp3._optimizer is opt2.fused and all(
            torch.allclose(p.grad, refs[0][i] + refs[1][i], atol=1e-6) for i, p in enumerate(net.parameters()))
        q.put((rank, rebound, old_detached, in_new_arena, ok_sum, ok_factor, ok_late, None))
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_optimizer_rebuilt_over_one_wrapper_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rebuild_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[-1] is None and all(r[1:-1]), r


def _inorder8_worker(rank, world, port, q):
    """eight ranks, every rank 'drops' a different pair of parameters (encoder_layerdrop draws diverge per rank): the
    all-reduces must still be ISSUED in bucket order 0, 1, 2, ... on every rank and the sums must be right"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from unispeech_amd.dp import GradReducer
        torch.manual_seed(0)
        params = [torch.nn.Parameter(torch.zeros(24 + 8 * (i % 3))) for i in range(24)]
        offsets, off = [], 0
        for p in params:
            offsets.append(off)
            off += (p.numel() + 7) // 8 * 8
        flat = torch.zeros(off)
        for p, o in zip(params, offsets):
            p.grad = flat[o:o + p.numel()].view_as(p)
        red = GradReducer(params, flat, offsets, bucket_bytes=256)
        issued = []
        orig = red._launch
        red._launch = lambda b: (issued.append(b), orig(b))[1]
        order = sorted(range(len(params)), key=lambda i: -offsets[i])
        drop = {order[(3 * rank + 1) % len(order)], order[(5 * rank + 7) % len(order)]}
        for i in order:
            if i not in drop:
                params[i].grad.add_(float(rank + 1))
                red._mark(i)
        mid = list(issued)
        red.finish()
        ok_order = issued == list(range(len(red.buckets))) and mid == list(range(len(mid)))
        want = torch.zeros_like(flat)
        for r_ in range(world):
            d_ = {order[(3 * r_ + 1) % len(order)], order[(5 * r_ + 7) % len(order)]}
            for i in range(len(params)):
                if i not in d_:
                    want[offsets[i]:offsets[i] + params[i].numel()] += float(r_ + 1)
        live = torch.zeros_like(flat, dtype=torch.bool)
        for p_, o_ in zip(params, offsets):
            live[o_:o_ + p_.numel()] = True
        q.put((rank, ok_order, bool(torch.allclose(flat[live], want[live])), len(red.buckets) >= 6, None))
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_in_order_issue_with_divergent_layerdrop_gloo_world8():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_inorder8_worker, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[-1] is None and all(r[1:-1]), r


class _RecordingTransport:
    """stands where dp.NativeTransport (the C-ABI RCCL reducer) stands: records the protocol GradReducer drives and performs the
    all-reduce through torch.distributed so that the results can be compared with the default transport"""

    def __init__(self):
        self.calls = []
        self.closed = False

    def bucket_ready(self, view):
        self.calls.append(("ready", view.data_ptr(), view.numel()))
        dist.all_reduce(view, op=dist.ReduceOp.SUM)

    def finish(self):
        self.calls.append(("finish",))

    def close(self):
        self.closed = True


def _worker_transport(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unispeech_amd.dp import GradReducer
    net, params, offsets, flat = _build(0)
    tr = _RecordingTransport()
    red = GradReducer(params, flat, offsets, bucket_bytes=1024, transport=tr)
    assert red.comm_stream is None and red.transport is tr
    x = torch.randn(4, 16, generator=torch.Generator().manual_seed(100 + rank))
    refs = []
    for r in range(world):
        n2, p2, o2, f2 = _build(0)
        xr = torch.randn(4, 16, generator=torch.Generator().manual_seed(100 + r))
        n2(xr).pow(2).sum().backward()
        refs.append(f2.clone())
    net(x).pow(2).sum().backward()
    red.finish()
    ok_sum = torch.allclose(flat, refs[0] + refs[1], atol=1e-6)
    ready = [c for c in tr.calls if c[0] == "ready"]
    # every bucket exactly once, in bucket-index order (the order RCCL matches collectives by), each the bucket's arena range;
    # one finish behind them
    esz = flat.element_size()
    want = [("ready", flat.data_ptr() + b["lo"] * esz, b["hi"] - b["lo"]) for b in red.buckets]
    ok_proto = ready == want and tr.calls[-1] == ("finish",) and sum(c[0] == "finish" for c in tr.calls) == 1
    # the sequence is the same on every rank
    seqs = [None] * world
    dist.all_gather_object(seqs, [(c[0], c[2] if len(c) > 2 else 0) for c in tr.calls])
    ok_same = all(s == seqs[0] for s in seqs)
    # accumulation step: nothing goes out under no_sync
    n0 = len(tr.calls)
    flat.zero_()
    with red.no_sync():
        net(x).pow(2).sum().backward()
    ok_nosync = len(tr.calls) == n0
    net(x).pow(2).sum().backward()
    red.finish()
    ok_acc = torch.allclose(flat, 2 * (refs[0] + refs[1]), atol=1e-5)
    red.close()
    q.put((rank, ok_sum, ok_proto, ok_same, ok_nosync, ok_acc, tr.closed))
    dist.destroy_process_group()


def test_reducer_drives_a_native_transport_with_the_bucket_protocol():
    """GradReducer with a transport object (the seat of the C-ABI RCCL reducer, dp.NativeTransport / WAVLM_DP_NATIVE=1): every
    bucket is reported once, in index order, as its arena range; one finish per step; no_sync holds everything back; close()
    reaches the transport; results equal the torch transport's."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_transport, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert all(r[1:]), r
