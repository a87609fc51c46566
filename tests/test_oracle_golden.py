"""CPU: the oracle restatement is pinned to outputs of the reference's own (importable) modules.

Fixtures in tests/golden/ were produced by oracle/make_golden.py executing reference
src/models/attention_processor.py and src/models/resampler.py unmodified.
"""
import os

import numpy as np
import pytest
import torch

from oracle.attention_ref import ip_region_mask, mask_grid_size, masked_ip_cross_attention, self_attention
from oracle.resampler_ref import resampler_forward


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_region_masks_bit_exact(golden_dir):
    g = _load(golden_dir, "ip_region_masks.npz")
    names = sorted({k[: -len("_bbox")] for k in g.files if k.endswith("_bbox")})
    assert len(names) >= 6
    for n in names:
        bbox = torch.tensor(g[n + "_bbox"])
        h, w = (int(v) for v in g[n + "_hw"])
        m = ip_region_mask(bbox, h * w, 1, h / w, 64, 16)
        masked = (m[:, 0] < -1).numpy().astype(np.uint8)
        assert masked.shape == g[n + "_masked"].shape
        assert (masked == g[n + "_masked"]).all(), n
        assert mask_grid_size(h * w, h / w) == (h, w)


def test_masked_ip_processor_matches_reference(golden_dir):
    g = _load(golden_dir, "masked_ip_attn.npz")
    T = lambda k: torch.tensor(g[k])
    h, w = (int(v) for v in g["hw"])
    y = masked_ip_cross_attention(T("x"), T("enc"), T("bbox"), h / w, T("wq"), T("wk"), T("wv"), T("wk_ip"), T("wv_ip"),
                                  T("wo"), T("bo"), int(g["heads"]), float(g["scale"]), 64, 16)
    assert torch.allclose(y, T("y"), atol=2e-6, rtol=1e-5)


def test_self_attn_processor_matches_reference(golden_dir):
    g = _load(golden_dir, "self_attn.npz")
    T = lambda k: torch.tensor(g[k])
    y = self_attention(T("x"), T("wq"), T("wk"), T("wv"), T("wo"), T("bo"), int(g["heads"]))
    assert torch.allclose(y, T("y"), atol=2e-6, rtol=1e-5)


@pytest.mark.parametrize("key_in,key_out", [("in", "out"), ("zero", "out_zero")])
def test_resampler_matches_reference(golden_dir, key_in, key_out):
    g = _load(golden_dir, "resampler.npz")
    sd = {k[3:]: torch.tensor(g[k]) for k in g.files if k.startswith("sd.")}
    x, m = torch.tensor(g["in_x"]), torch.tensor(g["in_magi"])
    if key_in == "zero":
        x, m = torch.zeros_like(x), torch.zeros_like(m)
    y = resampler_forward(sd, x, m, 2, 64)
    assert y.shape == (1, 16 + 4 * 16, 256)
    assert torch.allclose(y, torch.tensor(g[key_out]), atol=1e-5, rtol=1e-5)
