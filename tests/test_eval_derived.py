"""functional.eval_derived: what inference keeps between calls (packed q|k|v, GEMM images of conv / pos_conv weights) and when
it is rebuilt.  Host logic only: CPU tensors, no library call."""
import torch

import unispeech_amd.functional as F


def test_cache_follows_every_way_a_parameter_changes():
    a, b = torch.nn.Parameter(torch.randn(4, 3)), torch.nn.Parameter(torch.randn(4))
    built = []

    def build():
        built.append(1)
        return (a.detach() * 2, b.detach() + 1)

    old_setting = F.set_eval_cache(True)                 # opt-in since round 6 (see test_cache_is_opt_in... below)
    r0 = F.eval_derived([a, b], "t", build)              # autograd on: never kept
    r1 = F.eval_derived([a, b], "t", build)
    assert len(built) == 2 and r0[0] is not r1[0]
    with torch.no_grad():
        c0 = F.eval_derived([a, b], "t", build)
        c1 = F.eval_derived([a, b], "t", build)
        assert len(built) == 3 and c1[0] is c0[0]
        other = F.eval_derived([a, b], "another tag", build)
        assert len(built) == 4 and other[0] is not c0[0]
        b.add_(1.0)                                      # in-place through torch: version counter
        c2 = F.eval_derived([a, b], "t", build)
        assert len(built) == 5 and torch.equal(c2[1], b.detach() + 1)
        F.PARAM_EPOCH[0] += 1                            # a writer behind torch's back (optim.FusedAdam.step)
        F.eval_derived([a, b], "t", build)
        assert len(built) == 6
        a.data = a.data.clone()                          # moved (an optimizer arena took the parameter in)
        F.eval_derived([a, b], "t", build)
        assert len(built) == 7
        F.eval_derived([a, b], "t", build)
        assert len(built) == 7
    # inside a Function.forward grad mode is always off: the caller decides
    with torch.no_grad():
        F.eval_derived([a, b], "t", build, inference=False)
        assert len(built) == 8
    F.set_eval_cache(False)
    try:
        with torch.no_grad():
            F.eval_derived([a, b], "t", build)
            assert len(built) == 9
    finally:
        F.set_eval_cache(old_setting)


def test_cache_is_opt_in_because_data_writes_are_invisible():
    """the reference's optimizers update through `p.data` (optim/adam.py:172-226, fp16_optimizer.py:155-165),
    which moves neither the parameter's version counter nor its address.  By DEFAULT nothing is kept, so such an update is always
    seen; with the cache opted into, every writer of this package (and the model's train() / eval() / load_state_dict) calls
    invalidate_derived(), and a foreign `.data` writer has to."""
    import os
    assert os.environ.get("WAVLM_EVAL_CACHE", "0") != "1" and F.EVAL_CACHE is False    # the default
    p = torch.nn.Parameter(torch.ones(3))
    with torch.no_grad():
        v0 = F.eval_derived([p], "d", lambda: p.detach() * 2)
        ver = p._version
        p.data.add_(1.0)                                  # how fairseq's Adam writes
        assert p._version == ver                          # ... and why a version-counter key cannot see it
        v1 = F.eval_derived([p], "d", lambda: p.detach() * 2)
        assert torch.equal(v1, torch.full((3,), 4.0)) and not torch.equal(v0, v1)   # default: rebuilt, fresh
        with F.frozen_parameters():
            k0 = F.eval_derived([p], "d", lambda: p.detach() * 2)
            assert F.eval_derived([p], "d", lambda: p.detach() * 2) is k0           # kept
            p.data.add_(1.0)
            F.invalidate_derived()                        # the writer's duty under the opt-in
            k1 = F.eval_derived([p], "d", lambda: p.detach() * 2)
            assert torch.equal(k1, torch.full((3,), 6.0))
        assert F.EVAL_CACHE is False


def test_model_transitions_and_arena_writers_invalidate():
    """train() / eval() / load_state_dict of the modules that own derived tensors, and FusedAdam's arena-level writers, all bump
    the epoch (ADVICE r5 high + medium); host logic only"""
    from unispeech_amd.wavlm import WavLM, WavLMConfig
    from conftest import TINY
    m = WavLM(WavLMConfig(dict(TINY)))
    e = F.PARAM_EPOCH[0]
    m.eval()
    assert F.PARAM_EPOCH[0] > e
    e = F.PARAM_EPOCH[0]
    m.train()
    assert F.PARAM_EPOCH[0] > e
    e = F.PARAM_EPOCH[0]
    m.load_state_dict(m.state_dict())
    assert F.PARAM_EPOCH[0] > e
    from unispeech_amd.optim import FusedAdam
    e = F.PARAM_EPOCH[0]
    FusedAdam._params_written()
    assert F.PARAM_EPOCH[0] == e + 1
    import inspect
    src = inspect.getsource(FusedAdam.load_state_dict)
    assert "_params_written" in src                       # the arena copy at the end of load_state_dict tells inference
    assert "_params_written" in inspect.getsource(FusedAdam.step)


def test_inference_flag_is_thread_local():
    """a no_grad evaluation thread beside a training thread must not make the training forward skip the stores
    its backward reads"""
    import threading
    seen = {}
    F._INFERENCE_CALL[0] = True
    try:
        t = threading.Thread(target=lambda: seen.setdefault("other", F._INFERENCE_CALL[0]))
        t.start(); t.join()
        assert seen["other"] is False and F._INFERENCE_CALL[0] is True
    finally:
        F._INFERENCE_CALL[0] = False


def test_entries_die_with_the_parameter():
    p = torch.nn.Parameter(torch.randn(3))
    with torch.no_grad(), F.frozen_parameters():
        F.eval_derived([p], "x", lambda: p.detach() + 1)
    k = id(p)
    assert k in F._EVAL_DERIVED
    del p
    import gc
    gc.collect()
    assert k not in F._EVAL_DERIVED


def test_conv_grad_pad_is_the_geometry_conv_backward_uses():
    for (T_in, k, s) in [(101, 3, 2), (64, 2, 2), (37, 3, 2), (200, 2, 2), (1000, 10, 5)]:
        _, _, fp, bp = F._conv_geometry(T_in, k, s)
        assert F.conv_grad_pad(T_in, k, s) == (fp, bp)


def test_a_function_forward_learns_from_infer_apply_whether_the_call_is_inference():
    """inside Function.forward grad mode is always off and needs_input_grad is True for parameters even under no_grad"""
    seen = []

    class Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            seen.append((F._INFERENCE_CALL[0], torch.is_grad_enabled(), ctx.needs_input_grad[1]))
            return x * w

        @staticmethod
        def backward(ctx, g):
            return None, None

    x, w = torch.randn(3), torch.nn.Parameter(torch.randn(3))
    F.infer_apply(Fn, x, w)
    with torch.no_grad():
        F.infer_apply(Fn, x, w)
        Fn.apply(x, w)                                   # a direct apply never claims inference
    assert seen == [(False, False, True), (True, False, True), (False, False, True)]
    assert F._INFERENCE_CALL[0] is False
