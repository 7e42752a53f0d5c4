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
