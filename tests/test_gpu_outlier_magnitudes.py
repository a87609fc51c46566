"""GPU: one SDXL-shape UNet forward in the activation regime of a TRAINED checkpoint (missing 1, weak 3, next 3b).

Every other UNet test uses fan-in-normal random weights, which keep hidden states O(1).  Here the same seeded weights are
rescaled (tests/_outliers.py) so that, like a trained SDXL,
  * the transformer residual streams carry outlier channels of |h| ~ 10^2 - 10^3 in ~1 % of their channels on every token
    (the LayerNorm gains suppress them and lift the rest), and
  * self-attention logits reach +-30 and beyond,
and the HIP launch plan is compared with `UNetOracle(q = fp16 storage)` on identical weights and inputs - reference path
src/models/unet.py:116-347, attention_processor.py:76-78.  What this exercises at MODEL level for the first time:
  * the LayerNorm folded into the GEMM pair around it: statistics of raw rows whose sigma is set by the outliers, the rank-1
    mean correction in f16 (hi, lo) pairs (csrc/gemm_pp.hip FUSE 1 / 4 / 9, csrc/gemm.hip LNF) - batch 2 runs the 128-wide
    kernels' form, the batch-64 replica the 256 x 256 ping-pong kernel's;
  * `self_attn_sp_kernel`'s re-centring branch (no running maximum; a row is re-centred only when a partial sum of f16
    probabilities crosses 2^14): the test ASSERTS through ds_debug_counter("attn_sp_recentre") that the branch ran;
  * packed-f16 residual adds on streams whose ulp is 0.25-0.5.
Gates: finite; rel-L2 <= 5e-3 vs the fp16-storage oracle (the tolerance of the O(1) regime); the regime itself is measured on
the oracle side (largest |h| entering a LayerNorm, share of channels above 100, largest logit) and asserted.
"""
import pytest
import torch

from tests._gates import gate

pytestmark = pytest.mark.gpu
DEV = "cuda"
hq = lambda t: t.half().float()


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-6)).item()


def _counter(lib, reset):
    import ctypes as C
    v = C.c_longlong(0)
    assert lib.ds_debug_counter(b"attn_sp_recentre", int(reset), C.byref(v)) == 0
    return int(v.value)


def test_unet_sdxl_forward_with_outlier_channels_and_sharp_attention(hip_lib):
    from diffsensei_amd import _lib
    from diffsensei_amd.unet import UNetMangaModel
    from diffsensei_amd.unet_config import random_state_dict, sdxl_config
    from oracle.unet_ref import UNetOracle
    from tests._outliers import RegimeProbe, make_outlier_state_dict
    from tests.test_gpu_unet import _inputs
    lib = _lib.load()
    cfg = sdxl_config()
    base = random_state_dict(cfg, 0, DEV, torch.float16)
    sd = make_outlier_state_dict(base, amp=300.0, logit_sigma=3.5)       # fp16-representable fp32, CPU
    del base
    m = UNetMangaModel(cfg, device=DEV)
    m.load_state_dict({k: v.half() for k, v in sd.items()})
    m._attn_processors = {"x": type("P", (), {"scale": 0.6})()}
    x, enc, te, tid, bbox, db = _inputs(cfg, 2, 128, 128, seed=29)
    kw = lambda bb, t_e, t_i, d: dict(cross_attention_kwargs={"bbox": bb, "aspect_ratio": 1.0},
                                      added_cond_kwargs={"text_embeds": t_e, "time_ids": t_i}, dialog_bbox=d)
    _counter(lib, True)
    y = m(x.to(DEV), 801.0, enc.to(DEV), **kw(bbox, te, tid, db)).sample
    n_recentre_b2 = _counter(lib, True)
    assert y.shape == (2, 4, 128, 128) and torch.isfinite(y).all()
    # ---- the benched dispatch (UNet batch 64: gemm_pp_kernel with its fused LayerNorms, conv_halo256, the N = 4096 grids)
    rep64 = lambda t: torch.cat([t[:1].repeat(32, *([1] * (t.dim() - 1))), t[1:].repeat(32, *([1] * (t.dim() - 1)))])
    y64 = m(rep64(x).to(DEV), 801.0, rep64(enc).to(DEV), **kw(rep64(bbox), rep64(te), rep64(tid), rep64(db))).sample
    n_recentre_b64 = _counter(lib, True)
    assert torch.isfinite(y64).all()
    for r in range(64):
        assert torch.equal(y64[r], y64[0 if r < 32 else 32]), f"row {r} of the batch-64 forward differs from its replica"
    y64 = torch.stack([y64[0], y64[32]]).clone()
    eng64 = m._engines[next(k for k in m._engines if k[0] == 64)]
    print(f"re-centring branch of self_attn_sp_kernel: {n_recentre_b2} times in the batch-2 forward, {n_recentre_b64} in the "
          f"batch-64 forward; batch-64 plan: {getattr(eng64, 'ln_fused_blocks', 0)} transformer blocks with fused LayerNorms")
    assert n_recentre_b2 > 0 and n_recentre_b64 > 0, "the inputs did not drive self_attn_sp_kernel's re-centring branch"
    assert getattr(eng64, "ln_fused_blocks", 0) == 70
    del m
    torch.cuda.empty_cache()
    # ---- oracle, with the regime measured on its side
    with torch.no_grad():
        o16 = UNetOracle(cfg, sd, q=hq)
        o16.ip_scale = 0.6
        probe = RegimeProbe(o16)
        try:
            r16 = o16.forward(x, 801.0, enc, te, tid, bbox, 1.0, db)
        finally:
            probe.close()
    print(f"regime (oracle side): max |h| entering a LayerNorm {probe.max_h:.0f}, share of channels above 100: "
          f"{probe.frac_big:.4f}, largest self-attention logit {probe.max_logit:.1f}; output std {float(r16.std()):.3f}")
    assert torch.isfinite(r16).all()
    assert 100.0 <= probe.max_h <= 4000.0 and 0.005 <= probe.frac_big <= 0.03, (probe.max_h, probe.frac_big)
    assert probe.max_logit >= 30.0, probe.max_logit
    gate("outlier-regime SDXL UNet 1024x1024 batch 2 vs fp16-storage oracle", _rel(y, r16), 5e-3)
    gate("outlier-regime SDXL UNet 1024x1024 rows of batch 64 vs fp16-storage oracle", _rel(y64, r16), 5e-3)
    gate("outlier-regime rows of batch 64 vs batch 2", max(_rel(y64[0], y[0]), _rel(y64[1], y[1])), 4e-3)
    assert _rel(y[1], y[0]) > 1e-3
