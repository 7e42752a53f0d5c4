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
    old = F.EVAL_CACHE
    F.EVAL_CACHE = False
    try:
        with torch.no_grad():
            F.eval_derived([a, b], "t", build)
            assert len(built) == 9
    finally:
        F.EVAL_CACHE = old


def test_entries_die_with_the_parameter():
    p = torch.nn.Parameter(torch.randn(3))
    with torch.no_grad():
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
