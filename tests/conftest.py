import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def golden_state_dict(z, as_param=False):
    sd = {}
    for k in z.files:
        if k.startswith("sd/"):
            t = torch.from_numpy(np.array(z[k]))
            sd[k[3:]] = t.requires_grad_(True) if (as_param and t.is_floating_point()) else t
    return sd


class Cfg:
    """attribute bag with the reference's config field names"""

    def __init__(self, **kw):
        self.__dict__.update(kw)


TINY = dict(
    extractor_mode="default", encoder_layers=2, encoder_embed_dim=64, encoder_ffn_embed_dim=128,
    encoder_attention_heads=2, activation_fn="gelu", layer_norm_first=False,
    conv_feature_layers="[(32,10,5)] + [(32,3,2)] * 4 + [(32,2,2)] * 2", conv_bias=False, feature_grad_mult=0.1,
    dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, encoder_layerdrop=0.0, dropout_input=0.0,
    dropout_features=0.0, mask_length=4, mask_prob=0.65, mask_selection="static", mask_other=0,
    no_mask_overlap=False, mask_min_space=1, mask_channel_prob=0.0, conv_pos=16, conv_pos_groups=4,
    relative_position_embedding=True, num_buckets=32, max_distance=64, gru_rel_pos=True,
    label_rate=50, final_dim=32, logit_temp=0.1, skip_masked=False, skip_nomask=False, untie_final_proj=False,
    target_glu=False, boundary_mask=False, expand_attention_head_size=-1, normalize=False,
    mask_channel_length=10, mask_channel_selection="static", mask_channel_other=0, no_mask_channel_overlap=False,
    mask_channel_min_space=1,
)


@pytest.fixture
def tiny_cfg():
    return Cfg(**TINY)
