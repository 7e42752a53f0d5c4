"""WavLM-Large structure (BASELINE.json configs[3]: 24 layers, d = 1024, FFN 4096, 16 heads, extractor_mode "layer_norm",
pre-LN blocks) through the same step as bench.py: forward + masked-prediction loss + backward + fused Adam, bf16,
synthetic 15 s utterances.  Not the headline metric -- a data point that the path runs and how fast at Large shapes.
usage: python tools/bench_large.py [batch] [steps]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unispeech_amd.optim import FusedAdam  # noqa: E402
from unispeech_amd.pretrain import WavLMCriterion, WavLMPretrainConfig, WavLMPretrainModel  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
V, SECONDS, SR = 504, 15.0, 16000
cfg = WavLMPretrainConfig(
    encoder_layers=24, encoder_embed_dim=1024, encoder_ffn_embed_dim=4096, encoder_attention_heads=16,
    layer_norm_first=True, extractor_mode="layer_norm", dropout=0.0, attention_dropout=0.0, activation_dropout=0.0,
    encoder_layerdrop=0.0, dropout_input=0.1, dropout_features=0.1, feature_grad_mult=1.0, mask_prob=0.80, mask_length=10,
    final_dim=768, logit_temp=0.1, relative_position_embedding=True, num_buckets=320, max_distance=800, gru_rel_pos=True,
    label_rate=50, conv_feature_layers="[(512,10,5)] + [(512,3,2)] * 4 + [(512,2,2)] * 2")
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = WavLMPretrainModel(cfg, None, [range(V)]).to(dev).to(torch.bfloat16).train()
opt = FusedAdam(model.parameters(), lr=1e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.01, clip_norm=10.0, model=model)
crit = WavLMCriterion(None, 1.0, 0.0, loss_weights=[10.0], defer_logging=True)
T = int(SECONDS * SR)
g = torch.Generator().manual_seed(1234)
wav = torch.randn(B, T, generator=g).to(dev).to(torch.bfloat16)
pm_cpu = torch.zeros(B, T, dtype=torch.bool)
sample = {"id": torch.arange(B), "net_input": {"source": wav, "padding_mask": pm_cpu.to(dev), "padding_mask_cpu": pm_cpu},
          "target_list": [torch.randint(4, V, (B, int(50 * SECONDS)), generator=g).to(dev)]}
np.random.seed(1337)


def step():
    opt.zero_grad()
    loss, ss, _ = crit(model, sample)
    loss.backward()
    opt.step(grad_mult=1.0 / max(float(ss), 1.0))
    return loss


for _ in range(3):
    loss = step()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(steps):
    loss = step()
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
nparam = sum(p.numel() for p in model.parameters())
print("WavLM-Large structure: %.1f M parameters, batch %d x 15 s: %.1f ms/step, %.0f audio-s/s, loss %.1f"
      % (nparam / 1e6, B, dt * 1e3, B * SECONDS / dt, float(loss)))
