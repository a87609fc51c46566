"""Host side of the fused LayerNorm (diffsensei_amd/engine.py::pack_ln_fused, pack_geglu): the algebra the kernels implement,
checked in float64 on the CPU against torch's layer_norm + linear - the three `nn.LayerNorm`s of diffusers'
BasicTransformerBlock [3P] that the reference reaches from /root/reference/src/models/unet.py:244-338.

    LN(x) W^T + b  =  rstd (x (gamma (.) W)^T - mean c) + b',     c_n = sum_k (gamma (.) W)_nk,   b' = b + W beta

What is pinned here: the packed operands (f16 gw, the negated (hi, lo) f16 pair of c summed from the ROUNDED gw, f16 b'), the
GEGLU row permutation applied to all three alike, the (-c hi, -c lo, b' hi, b' lo) rows of the operand-swapped form, and the
statistics format (per 64-column strip partial sums, four interleaved chains combined as (0 + 1) + (2 + 3)).  The kernels
themselves are tested on the GPU (tests/test_gpu_ln_fusion.py)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from diffsensei_amd.engine import pack_geglu, pack_ln_fused


def _case(M, N, K, seed, offset=0.3):
    g = torch.Generator().manual_seed(seed)
    x = ((torch.randn((M, K), generator=g) + offset) * (1.0 + torch.rand((M, 1), generator=g) * 3)).half()
    w = (torch.randn((N, K), generator=g) / math.sqrt(K)).half()
    b = (torch.randn((N,), generator=g) * 0.3).half()
    gamma = (1 + 0.2 * torch.randn(K, generator=g)).half()
    beta = (torch.randn(K, generator=g) * 0.2).half()
    return x, w, b, gamma, beta


def _stats_like_the_kernels(x):
    """(mean, rstd) from per-strip partial sums in the kernels' order: strips of 64 columns, chains q = strip mod 4 summed in
    strip order, then (0 + 1) + (2 + 3); float32 throughout."""
    xs = x.float().numpy().reshape(x.shape[0], -1, 64)
    s, q = xs.sum(-1, dtype=np.float32), (xs * xs).sum(-1, dtype=np.float32)
    ch_s, ch_q = np.zeros((4, x.shape[0]), np.float32), np.zeros((4, x.shape[0]), np.float32)
    for j in range(s.shape[1]):
        ch_s[j & 3] += s[:, j]
        ch_q[j & 3] += q[:, j]
    inv = np.float32(1.0 / x.shape[1])
    mean = ((ch_s[0] + ch_s[1]) + (ch_s[2] + ch_s[3])) * inv
    var = np.maximum(((ch_q[0] + ch_q[1]) + (ch_q[2] + ch_q[3])) * inv - mean * mean, np.float32(0))
    return mean.astype(np.float64), (1.0 / np.sqrt(var.astype(np.float64) + 1e-5))


@pytest.mark.parametrize("M,N,K,offset", [(64, 128, 128, 0.3), (33, 256, 640, 0.0), (16, 64, 1280, 4.0)])
def test_row_form_equals_layernorm_linear(M, N, K, offset):
    x, w, b, gamma, beta = _case(M, N, K, 5 + M + N + K, offset)
    ref = F.linear(F.layer_norm(x.double(), (K,), gamma.double(), beta.double(), 1e-5), w.double(), b.double()).numpy()
    gw, c2, bp = pack_ln_fused(w, b, gamma, beta)
    assert gw.dtype == c2.dtype == bp.dtype == torch.float16 and tuple(c2.shape) == (N, 2)
    nc = c2.double().sum(1).numpy()                                      # -c as the kernels rebuild it: hi + lo
    np.testing.assert_allclose(nc, -gw.double().sum(1).numpy(), rtol=0, atol=2e-6 * np.abs(gw.double()).sum(1).max().item())
    mean, rstd = _stats_like_the_kernels(x)
    acc = x.double().numpy() @ gw.double().numpy().T
    got = rstd[:, None] * (acc + mean[:, None] * nc[None, :]) + bp.double().numpy()[None, :]
    err = np.abs(got - ref).max() / np.abs(ref).max()
    assert err <= 1.5e-3, err                                            # f16 rounding of gamma (.) W and of b', nothing else


def test_geglu_packing_permutes_all_three_operands_alike():
    C = 128
    x, w, b, gamma, beta = _case(48, 8 * C, C, 17)
    z = F.linear(F.layer_norm(x.double(), (C,), gamma.double(), beta.double(), 1e-5), w.double(), b.double())
    gw, c2, bp = pack_ln_fused(w, b, gamma, beta)
    gwp, bpp = pack_geglu(gw, bp)
    half = 4 * C
    c2p = torch.stack([c2[:half].reshape(-1, 64, 2), c2[half:].reshape(-1, 64, 2)], dim=1).reshape(-1, 2)
    mean, rstd = _stats_like_the_kernels(x)
    y = rstd[:, None] * (x.double().numpy() @ gwp.double().numpy().T + mean[:, None] * c2p.double().sum(1).numpy()[None, :]) \
        + bpp.double().numpy()[None, :]
    # packed layout: every 128 columns = 64 hidden values followed by their 64 gates
    y = y.reshape(48, -1, 2, 64)
    hidden, gate = y[:, :, 0, :].reshape(48, half), y[:, :, 1, :].reshape(48, half)
    zz = z.numpy()
    assert np.abs(hidden - zz[:, :half]).max() <= 2e-3 * np.abs(zz).max()
    assert np.abs(gate - zz[:, half:]).max() <= 2e-3 * np.abs(zz).max()


def test_operand_swapped_rows_carry_c_and_bias_pairs():
    """V^T = Wv LN(x)^T per image: statistics along the output columns, (-c hi, -c lo, b' hi, b' lo) per output row - the layout
    PackedUNet stores as `attn1.to_v.cb_ln`."""
    C, N = 128, 40
    x, wv, _, gamma, beta = _case(N, C, C, 29)
    ref = (wv.double() @ F.layer_norm(x.double(), (C,), gamma.double(), beta.double(), 1e-5).t()).numpy()
    gw, c2, _ = pack_ln_fused(wv, None, gamma, beta)
    bf = (wv.double() @ beta.double()).float()
    bh = bf.half()
    cb = torch.cat([c2, torch.stack([bh, (bf - bh.float()).half()], dim=1)], dim=1)
    assert tuple(cb.shape) == (C, 4)
    mean, rstd = _stats_like_the_kernels(x)
    nc, bp = cb[:, :2].double().sum(1).numpy(), cb[:, 2:].double().sum(1).numpy()
    acc = gw.double().numpy() @ x.double().numpy().T                     # [C, N]
    got = rstd[None, :] * (acc + nc[:, None] * mean[None, :]) + bp[:, None]
    assert np.abs(got - ref).max() <= 1.5e-3 * np.abs(ref).max()
