"""The one-call-per-block path (unispeech_amd/layerfn.py -> wavlm_encoder_layer_fwd / _bwd, csrc/layer.hip) against the
composed path (one autograd node per kernel, unispeech_amd/functional.py): both issue the same kernels with the same
arguments, so loss and EVERY parameter gradient must agree BIT FOR BIT -- with the recipe dropouts on (the seeds are drawn
in the same order), with key padding, with layerdrop (a dropped first block = no relative position table for anyone), post-LN
(Base) and pre-LN with fused residual adds (Large), and with the UniSpeech-SAT speaker tap that pulls the pending
feed-forward branch out of the chain.  Parity of the composed path with the oracle is what the other GPU tests establish
(test_bf16_e2e_gpu.py and test_large_e2e_gpu.py run THROUGH the fused path since it exists)."""
import numpy as np
import pytest
import torch

from conftest import TINY

pytestmark = pytest.mark.gpu

V = 60


def _cfg(**over):
    from unispeech_amd.pretrain import WavLMPretrainConfig
    c = dict(TINY)
    c.update(encoder_layers=3, encoder_embed_dim=128, encoder_ffn_embed_dim=256, encoder_attention_heads=2,
             conv_feature_layers="[(64,10,5)] + [(64,3,2)] * 4 + [(64,2,2)] * 2", conv_pos=16, conv_pos_groups=4,
             dropout=0.1, attention_dropout=0.1, dropout_input=0.1, final_dim=32)
    c.update(over)
    return WavLMPretrainConfig(**{k: v for k, v in c.items() if k in WavLMPretrainConfig.__dataclass_fields__})


def _run(cfg, fused, padded=False, sat=False, steps=2, B=3, T=16000):
    from unispeech_amd import functional as F
    from unispeech_amd import layerfn
    from unispeech_amd.optim import FusedAdam
    from unispeech_amd.pretrain import WavLMCriterion, WavLMPretrainModel
    layerfn.LAYER_FUSED = fused
    try:
        torch.manual_seed(0)
        model = WavLMPretrainModel(cfg, None, [range(V)]).cuda().to(torch.bfloat16).train()
        model.instance_sampling = "host"
        opt = FusedAdam(model.parameters(), lr=1e-3, clip_norm=10.0, model=model)
        crit = WavLMCriterion(None, 1.0, 0.0, loss_weights=[10.0, 10.0, 0.0] if sat else [10.0])
        g = torch.Generator().manual_seed(7)
        wav = torch.randn(B, T, generator=g).to(torch.bfloat16).cuda()
        pm = torch.zeros(B, T, dtype=torch.bool)
        if padded:
            pm[1, 12000:] = True
            pm[2, 9000:] = True
        target = torch.randint(4, V, (B, T // 320), generator=g).cuda()
        sample = {"id": torch.arange(B), "net_input": {"source": wav, "padding_mask": pm.cuda(), "padding_mask_cpu": pm},
                  "target_list": [target]}
        np.random.seed(11)
        torch.manual_seed(5)
        F._SEED_CTR[0] = 0
        out = []
        for _ in range(steps):
            opt.zero_grad()
            loss, ss, _ = crit(model, sample)
            loss.backward()
            torch.cuda.synchronize()
            out.append((loss.detach().float().cpu().clone(), opt.flat_grad.detach().float().cpu().clone()))
            opt.step(grad_mult=1.0 / max(ss, 1))
        names = [n for n, _ in model.named_parameters()]
        return out, names, model
    finally:
        layerfn.LAYER_FUSED = True


def _count_nodes(model_out_loss):
    seen, stack, names = set(), [model_out_loss.grad_fn], []
    while stack:
        f = stack.pop()
        if f is None or f in seen:
            continue
        seen.add(f)
        names.append(type(f).__name__)
        stack.extend(n for n, _ in f.next_functions)
    return names


@pytest.mark.parametrize("case", ["post_ln", "post_ln_padded", "post_ln_layerdrop", "pre_ln", "pre_ln_padded",
                                  "pre_ln_sat_tap", "no_relpos", "post_ln_padded_stored_p", "pre_ln_stored_p",
                                  "post_ln_padded_bits", "pre_ln_bits"])
def test_fused_block_equals_composed_path_bit_for_bit(case):
    over = {}
    # *_stored_p: the attention keeps its probabilities for backward (WAVLM_ATTN_STORE_P=1): the block call carries them in
    # its `saved` region (attn_store_p in the descriptor), the composed path in a tensor of its own -- same kernels either way
    # *_bits: the forward keeps its dropout decisions as bit words (WAVLM_ATTN_STORE_P=bits, attn_store_p = 2); no suffix: the
    # default -- nothing kept, the dQ kernel hands its dropout decisions to the dK/dV kernel
    stored = False
    for suffix, val in (("_stored_p", True), ("_bits", "bits")):
        if case.endswith(suffix):
            stored, case = val, case[:-len(suffix)]
    from unispeech_amd import functional as F_
    old_store = F_.ATTN_STORE_P
    F_.ATTN_STORE_P = stored
    try:
        _fused_equals_composed(case, over)
    finally:
        F_.ATTN_STORE_P = old_store


def _fused_equals_composed(case, over):
    padded = case.endswith("padded")
    if case.startswith("pre_ln"):
        over.update(layer_norm_first=True, extractor_mode="layer_norm")
    if case == "post_ln_layerdrop":
        over.update(encoder_layerdrop=0.5, encoder_layers=6)
    if case == "pre_ln_sat_tap":
        over.update(utterance_contrastive_loss=True, utterance_contrastive_layer=2, num_instances=0, cross_sample_instances=10)
    if case == "no_relpos":
        over.update(relative_position_embedding=False, gru_rel_pos=False)
    cfg = _cfg(**over)
    a, names, _ = _run(cfg, True, padded, sat=case == "pre_ln_sat_tap")
    b, _, _ = _run(cfg, False, padded, sat=case == "pre_ln_sat_tap")
    for step, ((la, ga), (lb, gb)) in enumerate(zip(a, b)):
        assert torch.equal(la, lb), "step %d: loss %r (one call per block) != %r (composed)" % (step, la.item(), lb.item())
        assert torch.isfinite(ga).all()
        assert ga.abs().max() > 0
        if not torch.equal(ga, gb):
            d = (ga - gb).abs()
            raise AssertionError("step %d: gradient arenas differ in %d of %d elements, max |diff| %.3e (max |g| %.3e)"
                                 % (step, int((d > 0).sum()), d.numel(), d.max().item(), gb.abs().max().item()))


@pytest.mark.parametrize("pre_ln", [False, True])
def test_fused_block_equals_composed_path_at_model_width(pre_ln):
    """Base / Large width with enough rows (8 x 2.5 s = 992 frames) for the grouped weight-gradient launches and the
    192 x 384 tiles: the C side regroups the four dW of a block exactly as functional.WgradGroup does"""
    over = dict(encoder_layers=2, conv_feature_layers="[(512,10,5)] + [(512,3,2)] * 4 + [(512,2,2)] * 2", conv_pos=128,
                conv_pos_groups=16, num_buckets=320, max_distance=800, final_dim=256)
    if pre_ln:
        over.update(encoder_embed_dim=1024, encoder_ffn_embed_dim=4096, encoder_attention_heads=16, layer_norm_first=True,
                    extractor_mode="layer_norm")
    else:
        over.update(encoder_embed_dim=768, encoder_ffn_embed_dim=3072, encoder_attention_heads=12)
    cfg = _cfg(**over)
    a, _, _ = _run(cfg, True, steps=1, B=8, T=40000)
    b, _, _ = _run(cfg, False, steps=1, B=8, T=40000)
    assert torch.equal(a[0][0], b[0][0]), (a[0][0].item(), b[0][0].item())
    d = (a[0][1] - b[0][1]).abs()
    assert torch.equal(a[0][1], b[0][1]), "gradient arenas differ in %d of %d elements, max |diff| %.3e" % (
        int((d > 0).sum()), d.numel(), d.max().item())


def test_fused_path_is_the_one_that_runs_and_is_one_node_per_block():
    """the benchmarked configuration takes the one-call path: one EncoderLayerFn node per executed block in the graph and
    none of the per-kernel nodes of an encoder block"""
    from unispeech_amd import layerfn
    from unispeech_amd.optim import FusedAdam
    from unispeech_amd.pretrain import WavLMCriterion, WavLMPretrainModel
    cfg = _cfg()
    torch.manual_seed(0)
    model = WavLMPretrainModel(cfg, None, [range(V)]).cuda().to(torch.bfloat16).train()
    opt = FusedAdam(model.parameters(), lr=1e-3, model=model)
    crit = WavLMCriterion(None, 1.0, 0.0, loss_weights=[10.0])
    B, T = 2, 16000
    wav = torch.randn(B, T).to(torch.bfloat16).cuda()
    pm = torch.zeros(B, T, dtype=torch.bool)
    sample = {"id": torch.arange(B), "net_input": {"source": wav, "padding_mask": pm.cuda(), "padding_mask_cpu": pm},
              "target_list": [torch.randint(4, V, (B, 50)).cuda()]}
    opt.zero_grad()
    loss, _, _ = crit(model, sample)
    nodes = _count_nodes(loss)
    assert sum(n.startswith("EncoderLayerFn") for n in nodes) == cfg.encoder_layers
    assert not any(n.startswith(("AttnCoreFn", "FFNFn", "GateFn")) for n in nodes)
    loss.backward()
    torch.cuda.synchronize()
    assert torch.isfinite(opt.flat_grad.float()).all()
    assert layerfn.LAYER_FUSED


def test_c_level_gradient_listener_reports_every_sink_of_a_step():
    """wavlm_dp_set_listener (include/wavlm_hip.h): a caller BELOW Python learns which slice of its gradient arena a backward
    kernel has just accumulated into.  One training step with a ctypes callback registered: every parameter's arena slice
    is covered by the reported (base, bytes) ranges, every report lies inside the arena, and the blocks report in backward
    order (last block first) -- what a reducer needs to start bucket 0 while backward is still running."""
    import ctypes
    from unispeech_amd import _lib
    from unispeech_amd.optim import FusedAdam
    from unispeech_amd.pretrain import WavLMCriterion, WavLMPretrainModel
    cfg = _cfg()
    torch.manual_seed(0)
    model = WavLMPretrainModel(cfg, None, [range(V)]).cuda().to(torch.bfloat16).train()
    opt = FusedAdam(model.parameters(), lr=1e-3, model=model)
    crit = WavLMCriterion(None, 1.0, 0.0, loss_weights=[10.0])
    B, T = 2, 16000
    wav = torch.randn(B, T).to(torch.bfloat16).cuda()
    pm = torch.zeros(B, T, dtype=torch.bool)
    sample = {"id": torch.arange(B), "net_input": {"source": wav, "padding_mask": pm.cuda(), "padding_mask_cpu": pm},
              "target_list": [torch.randint(4, V, (B, 50)).cuda()]}
    got = []
    cb = _lib.GRAD_LISTENER(lambda base, nbytes, stream, user: got.append((int(base or 0), int(nbytes))))
    L = _lib.lib()
    opt.zero_grad()
    loss, _, _ = crit(model, sample)
    L.wavlm_dp_set_listener(ctypes.cast(cb, ctypes.c_void_p), None)
    try:
        loss.backward()
    finally:
        L.wavlm_dp_set_listener(None, None)
    torch.cuda.synchronize()
    lo, es = opt.flat_grad.data_ptr(), opt.flat_grad.element_size()
    hi = lo + opt.flat_grad.numel() * es
    inside = [(b, n) for b, n in got if lo <= b < hi]
    assert inside and all(b + n <= hi for b, n in inside)
    covered = torch.zeros(opt.flat_grad.numel(), dtype=torch.bool)
    for b, n in inside:
        covered[(b - lo) // es:(b - lo + n) // es] = True
    names = dict((id(p), n) for n, p in model.named_parameters())
    # what the C side sees: every gradient that a kernel ACCUMULATES into the arena.  Gradients that autograd adds itself
    # (tensors returned by a Function: conv0 / GroupNorm, pos_conv, mask_emb, label embeddings, the position embedding) are
    # the Python hooks' business (dp.GradReducer uses both).
    enc = [p for n, p in model.named_parameters() if n.startswith("encoder.layers.")
           and "relative_attention_bias" not in n]
    assert len(enc) >= 3 * 16
    for p, o in zip(opt.params, opt.offsets):
        if any(p is q for q in enc):
            assert bool(covered[o:o + p.numel()].all()), "no C-level notification covers " + names[id(p)]
    # order: the first encoder-block report belongs to the LAST block, the last one to block 0
    def block_of(addr):
        for p, o in zip(opt.params, opt.offsets):
            if lo + o * es <= addr < lo + (o + p.numel()) * es:
                n = names[id(p)]
                return int(n.split(".")[2]) if n.startswith("encoder.layers.") else None
        return None
    blocks = [b for b in (block_of(a) for a, _ in inside) if b is not None]
    assert blocks[0] == cfg.encoder_layers - 1 and blocks[-1] == 0
    assert blocks == sorted(blocks, reverse=True)
    # with the listener cleared nothing is reported
    got.clear()
    opt.zero_grad()
    loss, _, _ = crit(model, sample)
    loss.backward()
    torch.cuda.synchronize()
    assert not got


def test_deepcopy_and_pickle_after_a_fused_step():
    """the per-block bindings hold ctypes descriptors with pointer fields; once a fused step has run they must not
    be part of the module's state -- copy.deepcopy (EMA / teacher copies) and pickle of TRAINED encoder blocks have to work,
    and the copy builds its own bindings over its own parameters.  (The blocks, not the whole model: torch's weight_norm leaves
    a non-leaf `weight` on pos_conv after a training forward, which torch itself refuses to deep-copy -- as in the reference.)"""
    import copy
    import io
    import pickle
    from unispeech_amd import layerfn
    out, names, model = _run(_cfg(), True, steps=1)
    layers = model.encoder.layers
    assert layerfn._BINDINGS.get(layers[0]) is not None, "the fused path did not run (no binding was built)"
    clone = copy.deepcopy(layers)
    assert layerfn._BINDINGS.get(clone[0]) is None
    for (n, p), (_, q) in zip(layers.named_parameters(), clone.named_parameters()):
        assert p.data_ptr() != q.data_ptr() and torch.equal(p, q), n
    buf = io.BytesIO()
    pickle.dump(layers, buf)   # what torch.save(module) does
    buf.seek(0)
    again = pickle.load(buf)
    assert sum(p.numel() for p in again.parameters()) == sum(p.numel() for p in layers.parameters())
    assert not any("_wl_binding" in m.__dict__ for m in layers.modules())
