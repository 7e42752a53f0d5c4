"""The `--fp16` flag of the reference recipes on a part whose kernels compute in bf16.

Every shipped recipe passes --fp16 (src/examples/hubert/scripts/pretrain.sh:25): the reference Trainer then calls
`model.half()` / `criterion.half()` (trainer.py:86-89), casts the batch to fp16 (trainer.py:1141-1152) and wraps the
optimizer in FP16Optimizer with a DynamicLossScaler (optim/fp16_optimizer.py:186-289, optim/dynamic_loss_scaler.py:7-70).
The gfx950 kernels of this package have no fp16 instantiation: low precision here is bf16 (MFMA bf16, fp32 accumulate,
fp32 master weights in the optimizer).  Two behaviours, chosen explicitly:

  * default: `model.half()` and an fp16 optimizer build RAISE, naming this switch -- nobody gets bf16 arithmetic by accident;
  * `WAVLM_FP16_AS_BF16=1` (or `fairseq_plugin.register(..., fp16_as_bf16=True)`): an unmodified `--fp16` recipe runs.
    `model.half()` converts to bf16, the fp16 batch is cast to bf16 at the model's door, and the optimizer front-end keeps the
    reference's loss-scaling PROTOCOL around the bf16 kernels: the loss is multiplied by `scaler.loss_scale` before
    backward, 1 / loss_scale rides in the deferred gradient factor, a non-finite gradient norm raises OverflowError (the
    Trainer skips the update, trainer.py:856-862) and halves the scale, `scale_window` clean updates double it, and
    `optimizer.scaler.loss_scale` is there for the Trainer's logging (trainer.py:947) and travels in the optimizer's
    state dict (`loss_scale`, fp16_optimizer.py:79, 90-91).  Steady state: bf16 has fp32's exponent range, so the scale is
    not NEEDED -- but the reference's scaler doubles it every `scale_window` clean updates without a bound, so it climbs
    until the fp32 sum of squares of the scaled gradients overflows (scale x |g| ~ 2^64); from then on one update per window
    is skipped and the scale halved, exactly the rhythm of a real fp16 run, only tens of thousands of updates later.  The
    protocol is kept unmodified (no cap) so that checkpoints, logs and the overflow path of a recipe behave as they do in
    the reference.  Cost: with a scaler the gradient-norm check reads the norm on the host every update
    (`FairseqFusedAdam.clip_grad_norm`, as fp16_optimizer.py:199-206 does) -- one stream synchronisation per step that the
    --bf16 path does not have; the launch thread loses its run-ahead there (launch-thread section).
"""
import os

_STATE = {"fp16_as_bf16": os.environ.get("WAVLM_FP16_AS_BF16", "0") == "1"}

MESSAGE = ("unispeech_amd: fp16 is not a compute type of the MI355X path (bf16 / fp32 are).  Run the recipe with --bf16 "
           "(cfg.common.bf16=True, INTEGRATION.md section 1), or set WAVLM_FP16_AS_BF16=1 / "
           "fairseq_plugin.register(..., fp16_as_bf16=True) to run an unmodified --fp16 recipe on bf16 kernels with the "
           "reference's dynamic loss-scaling protocol (unispeech_amd/precision.py)")


def fp16_as_bf16():
    return _STATE["fp16_as_bf16"]


def set_fp16_as_bf16(on):
    _STATE["fp16_as_bf16"] = bool(on)
