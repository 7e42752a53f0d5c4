"""world_size-2 check of the data-parallel path on the HIP kernels (two ranks sharing cuda:0, gloo rendezvous on
127.0.0.1): with gradient sinks the backward kernels accumulate straight into the FusedAdam arena and tell the
reducer which slices are complete, so buckets are all-reduced while backward is still running.  After
all_reduce_grads() the arena must hold the SUM over ranks of the gradients each rank produces alone (no_sync), for
every parameter including the packed q|k|v slices and the twice-used ILS heads."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, grouped):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        import numpy as np
        import unispeech_amd.functional as F
        F.WGRAD_GROUPING = grouped  # grouped: weight gradients are queued and written later than autograd's hooks fire
        from unispeech_amd.dp import DataParallelWavLM
        from unispeech_amd.optim import FusedAdam
        from unispeech_amd.pretrain import WavLMPretrainConfig, WavLMPretrainModel, WavLMCriterion
        from test_model_gpu import BASE

        d = dict(BASE)  # Base width, 3 layers, ILS heads on layers 2 and 3 (final_proj / label_embs used twice)
        d.update(encoder_layers=3, predict_layers="[2,3]")
        cfg = WavLMPretrainConfig(**{k: v for k, v in d.items() if k in WavLMPretrainConfig.__dataclass_fields__})
        torch.manual_seed(0)  # same weights on both ranks
        model = WavLMPretrainModel(cfg, None, [range(104)]).cuda().to(torch.bfloat16).train()
        crit = WavLMCriterion(None, 1.0, 0.0, loss_weights=[10.0])
        opt = FusedAdam(model.parameters(), model=model)
        dp = DataParallelWavLM(model, opt, bucket_bytes=4 << 20)
        assert len(dp.reducer.buckets) >= 3
        B, T = 2, 24000
        g = torch.Generator().manual_seed(100 + rank)  # different data per rank
        wav = torch.randn(B, T, generator=g).cuda().to(torch.bfloat16)
        target = torch.randint(4, 104, (B, 75), generator=g).cuda()
        sample = {"id": torch.arange(B), "net_input": {"source": wav, "padding_mask": torch.zeros(B, T, dtype=torch.bool).cuda()},
                  "target_list": [target]}

        def backward():
            np.random.seed(11 + rank)
            torch.manual_seed(5)
            F._SEED_CTR[0] = 0
            loss, _, _ = crit(dp, sample)
            loss.backward()

        opt.zero_grad()
        with dp.no_sync():
            backward()
        local = opt.flat_grad.detach().float().cpu().clone()
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        want = sum(gathered)

        opt.zero_grad()
        dp.reducer.record_trace = True
        dp.reducer.trace.clear()
        backward()
        end_bwd = torch.cuda.Event(enable_timing=True)
        end_bwd.record()                   # compute stream: everything backward enqueued lies before this event
        early = sum(dp.reducer._launched)  # buckets already in flight when backward returned
        dp.all_reduce_grads()
        torch.cuda.synchronize()
        # device-side evidence of the overlap: the point on the COMPUTE stream at which a bucket's all-reduce was handed to
        # the side stream (it waits for exactly that point) precedes the end of backward by this many milliseconds
        lead = [ev.elapsed_time(end_bwd) for _, ev in dp.reducer.trace[:early]]
        from unispeech_amd import ops as _ops
        assert _ops.get_reserved_cus() == 6, "persistent GEMM grids must leave CUs to the collectives (dp.GradReducer)"
        assert dp.reducer.comm_stream is not None and [b for b, _ in dp.reducer.trace] == list(range(len(dp.reducer.trace)))
        assert len(lead) == early and sum(1 for x in lead if x > 0.02) >= early // 2, lead
        assert abs(opt.pending_mult - 1.0 / world) < 1e-12   # the wrapper's average rides in the deferred factor
        got = opt.flat_grad.detach().float().cpu()
        scale = want.abs().max().clamp_min(1e-6)
        err = ((got - want).abs().max() / scale).item()
        nz = (local.abs() > 0).float().mean().item()
        q.put((rank, err, early, len(dp.reducer.buckets), nz, None))
    except Exception as e:  # surface the failure in the parent instead of a queue timeout
        import traceback
        q.put((rank, float("inf"), 0, 0, 0.0, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("grouped", [False, True])
def test_dp_world2_sink_gradients_sum_over_ranks(grouped):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, grouped)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, err, early, nb, nz, tb in res:
        assert tb is None, tb
        # bf16 arena: the reduced sum is rounded to bf16 (2^-9 relative) and the two backward passes differ by the order
        # of their float atomics; measured 3.3e-3 of the largest gradient.  A bucket reduced twice or too early shows
        # up as >= 5e-2.
        assert err < 1e-2, (rank, err)
        assert nz > 0.5, "gradient arena mostly empty"
        assert early >= nb // 2, f"only {early} of {nb} buckets were launched during backward"
        print(f"rank {rank}: {early}/{nb} buckets in flight at the end of backward, rel err {err:.2e}")


_NATIVE_SCRIPT = r"""
import ctypes, sys
import torch
sys.path.insert(0, %r)
from unispeech_amd import _lib, ops
L = _lib.lib()
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
ident = ctypes.create_string_buffer(128)
assert L.wavlm_dp_bucket_ready(None, 0, 1, None) == -1            # not initialised / bad argument: WL_EINVAL, no crash
assert L.wavlm_dp_finish(None) == -1
rc = L.wavlm_dp_unique_id(ident)
assert rc == 0, "RCCL could not be loaded (%%d)" %% rc
assert any(b != 0 for b in ident.raw)
assert L.wavlm_dp_init(0, 1, ident, 1) == 0                       # one rank, ncclAvg
assert L.wavlm_dp_init(0, 1, ident, 1) == -1                      # one communicator per process
# buckets of a "gradient arena" written on the compute stream, reported while later work is still queued there
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    arena = torch.zeros(3 << 20, device=dev, dtype=torch.bfloat16)
    spin = torch.randn(4096, 4096, device=dev)
    for _ in range(20):
        spin = spin @ spin * 1e-4                                   # keeps the compute stream busy in front of the writes
    arena[:1 << 20] = 1.0
    assert L.wavlm_dp_bucket_ready(ops.ptr(arena), 1 << 20, ops.dt(arena), ops.stream()) == 0
    arena[1 << 20:] = 2.0
    assert L.wavlm_dp_bucket_ready(ops.ptr(arena, 1 << 20), 2 << 20, ops.dt(arena), ops.stream()) == 0
    f32 = torch.full((1000,), 3.0, device=dev)
    assert L.wavlm_dp_bucket_ready(ops.ptr(f32), 1000, ops.dt(f32), ops.stream()) == 0
    assert L.wavlm_dp_finish(ops.stream()) == 0
    total = arena.float().sum() + f32.sum()                         # ordered behind the reductions by finish()
side.synchronize()
want = (1 << 20) * 1.0 + (2 << 20) * 2.0 + 3000.0
assert abs(float(total) - want) < 1e-3 * want, (float(total), want)   # world 1: sum / average = identity, in place
assert L.wavlm_dp_finish(ops.stream()) == 0                          # nothing pending: no-op
assert L.wavlm_dp_destroy() == 0
assert L.wavlm_dp_destroy() == 0
assert L.wavlm_dp_bucket_ready(ops.ptr(f32), 1000, ops.dt(f32), ops.stream()) == -1   # gone: WL_EINVAL again
print("NATIVE_DP_OK")
""" % (ROOT,)


def test_c_abi_rccl_reducer_single_rank():
    """wavlm_dp_unique_id / _init / _bucket_ready / _finish / _destroy on one GPU (an RCCL communicator of one rank): RCCL is
    resolved at run time, buckets are reduced in place on the library's stream behind the compute stream's writes, finish()
    orders the compute stream behind them.  In a process of its own (the communicator is process-wide)."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = subprocess.run([sys.executable, "-c", _NATIVE_SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "NATIVE_DP_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
