"""GPU: parity at the shapes of BASELINE.json configs[3]'s 1536 x 1536 bucket and configs[4] (2048 x 2048) - the sizes the
reference serves from its demo (scripts/demo/gradio_wo_mllm.py:243-244, README.md:33) and that round 4 benchmarked without a
test at their own shapes (missing 1 / weak 2, 3):

  * self-attention (reference: src/models/attention_processor.py:76-78) at (heads 10, N 16 384) and (heads 20, N 4 096) -
    2048 x 2048 - and (10, 9 216), (20, 2 304) - 1536 x 1536 -, every flash kernel the dispatch can pick there
    (`self_attn_kernel<1>`, `<2>`, `self_attn_sp_kernel`), against fp32
    softmax(Q K^T / 8) V evaluated on the device in row blocks (the score matrix of one head is 1 GiB at N = 16 384);
  * masked IP-Adapter attention (attention_processor.py:235-258) on 128 x 128, 96 x 96 and 64 x 64 x 20-head grids vs the oracle;
  * one whole UNet forward at 1536 x 1536 (192 x 192 latents, CFG batch 2, 4 character boxes, 2 dialog boxes) vs
    `UNetOracle(q = fp16 storage)`; the same at 2048 x 2048 behind DS_TEST_2048=1 (five minutes of host cores; its log
    is committed under profiles/);
  * the benched UNet batch with 64 DISTINCT items (seeds, boxes, dialog boxes, text embeddings all different): every row
    against the batch-2 forward of its own (unconditional, conditional) pair.

Tolerances are the ones of the smaller shapes: max |err| <= 3e-3 max|ref| on attention outputs, rel-L2 <= 5e-3 on a UNet
forward vs the fp16-storage oracle (measured 1.53-1.55e-3; the gate was 2e-2 until round 5), <= 4e-3 between two launch plans of the same inputs.
"""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
hq = lambda t: t.half().float()


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-6)).item()


def _close(got, ref, tol, what):
    got, ref = got.float().cpu(), ref.float().cpu()
    assert got.shape == ref.shape and torch.isfinite(got).all(), what
    err, den = (got - ref).abs().max().item(), max(ref.abs().max().item(), 1e-3)
    assert err <= tol * den + 1e-3 * tol, f"{what}: max err {err:.4g} vs max|ref| {den:.4g}"


def _sdpa_rows_on_device(q, k, v, heads, rows=2048):
    """fp32 softmax(q k^T / 8) v per head on the device, `rows` query rows at a time.  q, k, v: [1, N, heads*64] f16 (cuda)."""
    _, N, C = q.shape
    out = torch.empty((1, N, C), dtype=torch.float32, device=q.device)
    for h in range(heads):
        sl = slice(h * 64, (h + 1) * 64)
        kf, vf = k[0, :, sl].float(), v[0, :, sl].float()
        for i in range(0, N, rows):
            s = (q[0, i:i + rows, sl].float() @ kf.t()) * 0.125
            out[0, i:i + rows, sl] = torch.softmax(s, -1) @ vf
    return out.cpu()


@pytest.mark.parametrize("heads,N", [(10, 16384), (20, 4096), (10, 9216), (20, 2304)])
def test_self_attention_at_2048_and_1536_shapes(hip_lib, heads, N):
    """Q is scaled by 3 so the softmax has structure (logit sigma 3, a few hundred keys carry a row) instead of the
    near-uniform average 16 384 unit-variance keys give."""
    from diffsensei_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(heads * 100003 + N)
    C = heads * 64
    q = (torch.randn((1, N, C), generator=g) * 3.0).half().to(DEV)
    k = torch.randn((1, N, C), generator=g).half().to(DEV)
    v = torch.randn((1, N, C), generator=g).half().to(DEV)
    ref = _sdpa_rows_on_device(q, k, v, heads)
    vt = v.view(1, N, heads, 64).permute(0, 2, 3, 1).contiguous()
    got = {}
    try:
        for var in (0, 1, 2, 3):      # automatic dispatch, 32-row flash, 64-row flash, software-pipelined
            assert lib.ds_set_option(b"attn_variant", var) == 0
            got[var] = ops.self_attention(q, k, vt, heads).float().cpu()
    finally:
        lib.ds_set_option(b"attn_variant", 0)
    for var, y in got.items():
        _close(y, ref, 3e-3, f"self-attention variant {var} heads {heads} N {N}")
        r = _rel(y, ref)
        print(f"self-attention heads {heads} N {N} variant {var}: rel-L2 {r:.3e}")
        assert r <= 2e-3, (var, r)
    assert torch.equal(got[1], got[2]), "32-row and 64-row flash kernels differ"
    assert torch.equal(got[0], got[1]) or torch.equal(got[0], got[3]), "automatic dispatch ran none of the tested kernels"


@pytest.mark.parametrize("B,heads,hw", [(1, 10, (128, 128)), (2, 20, (64, 64)), (2, 10, (96, 96)), (1, 20, (48, 48)),
                                        (1, 10, (160, 96))])
def test_masked_ip_attention_at_2048_and_1536_grids(hip_lib, B, heads, hw):
    """`ip_attn_kernel` on the mask grids of 2048 x 2048 (128 x 128 at 640 channels, 64 x 64 at 1280), 1536 x 1536 (96 x 96,
    48 x 48) and a 2560 x 1536 portrait bucket, four boxes incl. overlapping ones, vs the oracle's masked SDPA
    (oracle/attention_ref.py, pinned to the reference's own masks in tests/golden)."""
    from diffsensei_amd import ops
    from diffsensei_amd.attention_processor import LP
    from oracle.attention_ref import _heads, ip_region_mask, sdpa
    _r = lambda shape, gen, scale=1.0: (torch.randn(shape, generator=gen) * scale).half()

    def _ip_attn_ref(q, enc, bbox, hw, wk, wv, wki, wvi, heads, scale):
        b, n, c = q.shape
        txt, ip = enc[:, :77], enc[:, 77:]
        qh = _heads(q.float(), heads)
        h_ = lambda t: _heads(t.half().float(), heads)
        t_out = sdpa(qh, h_(txt.float() @ wk.float().t()), h_(txt.float() @ wv.float().t()))
        msk = ip_region_mask(bbox, n, heads, hw[0] / hw[1], 64, 16)
        i_out = sdpa(qh, h_(ip.float() @ wki.float().t()), h_(ip.float() @ wvi.float().t()), msk)
        return (t_out + scale * i_out).transpose(1, 2).reshape(b, n, c)

    g = torch.Generator().manual_seed(B * 31 + heads + hw[0])
    N, C, X = hw[0] * hw[1], heads * 64, 128
    q, enc = _r((B, N, C), g), _r((B, 157, X), g)
    wk, wv, wki, wvi = (_r((C, X), g, 1 / math.sqrt(X)) for _ in range(4))
    bbox = torch.zeros(B, 4, 4)
    bbox[-1, 0] = torch.tensor([0.05, 0.10, 0.50, 0.95])
    bbox[-1, 1] = torch.tensor([0.50, 0.10, 0.95, 0.95])
    bbox[-1, 2] = torch.tensor([0.30, 0.30, 0.70, 0.60])
    bbox[-1, 3] = torch.tensor([0.00, 0.80, 1.00, 1.00])
    ref = _ip_attn_ref(q, enc, bbox, hw, wk, wv, wki, wvi, heads, 0.6)
    encd = enc.to(DEV)
    txt, ip = ops.pad_rows(encd, 0, 77, LP), ops.pad_rows(encd, 77, 80, LP)
    kt = ops.gemm(txt.view(-1, X), wk.to(DEV)).view(B, LP, C)
    ki = ops.gemm(ip.view(-1, X), wki.to(DEV)).view(B, LP, C)
    vtt, vti = ops.gemm_batched_nt(wv.to(DEV), txt), ops.gemm_batched_nt(wvi.to(DEV), ip)
    y = ops.masked_ip_attention(q.to(DEV), kt, vtt, ki, vti, bbox.to(DEV), heads, hw, 0.6)
    _close(y, ref, 4e-3, f"masked ip attention grid {hw}")
    assert _rel(y, ref) <= 3e-3


@pytest.fixture(scope="module")
def sdxl_model(hip_lib):
    from diffsensei_amd.unet import UNetMangaModel
    from diffsensei_amd.unet_config import sdxl_config
    cfg = sdxl_config()
    m = UNetMangaModel(cfg, device=DEV).init_random(0)
    m._attn_processors = {"x": type("P", (), {"scale": 0.6})()}
    return cfg, m


def _item(cfg, H, W, seed, cond, nbox=2):
    """One UNet batch item: latent, encoder states, pooled text, time ids, character boxes, dialog boxes - all from `seed`."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(1, 4, H, W, generator=g).half()
    enc = torch.randn(1, cfg.num_text_tokens + cfg.num_ip_tokens, cfg.cross_attention_dim, generator=g).half()
    te = torch.randn(1, cfg.projection_class_embeddings_input_dim - 6 * cfg.addition_time_embed_dim, generator=g).half()
    tid = torch.tensor([[H * 8, W * 8, 0, 0, H * 8, W * 8]], dtype=torch.float16)
    bbox, db = torch.zeros(1, 4, 4), torch.zeros(1, 8, 4, dtype=torch.float16)
    if cond:
        for j in range(nbox):
            x0, y0 = torch.rand(2, generator=g) * 0.5
            wd, ht = 0.2 + torch.rand(2, generator=g) * 0.3
            bbox[0, j] = torch.tensor([x0, y0, x0 + wd, y0 + ht])
        for j in range(1 + seed % 3):
            x0, y0 = torch.rand(2, generator=g) * 0.7
            db[0, j] = torch.tensor([x0, y0, x0 + 0.25, y0 + 0.12], dtype=torch.float16)
    return x, enc, te, tid, bbox, db


def _forward(m, items):
    x, enc, te, tid, bbox, db = (torch.cat(t) for t in zip(*items))
    return m(x.to(DEV), 801.0, enc.to(DEV), cross_attention_kwargs={"bbox": bbox, "aspect_ratio": 1.0},
             added_cond_kwargs={"text_embeds": te, "time_ids": tid}, dialog_bbox=db).sample


def test_unet_sdxl_batch64_of_distinct_items(sdxl_model):
    """The benchmark's UNet batch (num_samples 32 -> 64 rows, 1024 x 1024) with 64 DIFFERENT items - 32 seeds x
    {unconditional: no boxes, conditional: two character boxes + one to three dialog boxes}, different latents, encoder
    states and pooled text embeddings per item: every output row must reproduce the batch-2 forward of its own pair
    (<= 4e-3: the batch-64 plan fuses the LayerNorms and runs the large-problem kernels, the batch-2 plan is the one
    `test_unet_sdxl_forward_vs_oracle_1024` ties to the oracle).  A fault that mixes items of one CFG half - per-item box /
    text-embedding indexing, the V^T fold's tile -> batch item map, per-item GroupNorm / LayerNorm statistics slots - shows
    up here and is invisible to a batch of replicated rows."""
    cfg, m = sdxl_model
    unc = [_item(cfg, 128, 128, 1000 + s, False) for s in range(32)]
    con = [_item(cfg, 128, 128, 2000 + s, True) for s in range(32)]
    y64 = _forward(m, unc + con)
    assert y64.shape == (64, 4, 128, 128) and torch.isfinite(y64).all()
    worst = 0.0
    for s in range(32):
        y2 = _forward(m, [unc[s], con[s]])
        d0, d1 = _rel(y64[s], y2[0]), _rel(y64[32 + s], y2[1])
        worst = max(worst, d0, d1)
        assert d0 <= 4e-3 and d1 <= 4e-3, (s, d0, d1)
    # and the items really are different problems
    assert _rel(y64[1], y64[0]) > 0.5 and _rel(y64[33], y64[32]) > 0.5
    from tests._gates import gate
    gate("batch 64 of distinct items vs 32 batch-2 forwards, worst row", worst, 4e-3)


def _oracle_case(sdxl_model, H, W, seed):
    from oracle.unet_ref import UNetOracle
    cfg, m = sdxl_model
    items = [_item(cfg, H, W, seed, False), _item(cfg, H, W, seed + 1, True, nbox=4)]
    y = _forward(m, items).float().cpu()
    assert y.shape == (2, 4, H, W) and torch.isfinite(y).all()
    torch.cuda.empty_cache()
    sd = {k: v.float().cpu() for k, v in m.state_dict().items()}
    x, enc, te, tid, bbox, db = (torch.cat(t) for t in zip(*items))
    with torch.no_grad():
        o16 = UNetOracle(cfg, sd, q=hq)
        o16.ip_scale = 0.6
        r16 = o16.forward(x, 801.0, enc, te, tid, bbox, 1.0, db)
    e = _rel(y, r16)
    print(f"SDXL {H * 8} x {W * 8} forward (CFG batch 2, 4 boxes): rel-L2 vs fp16-storage oracle {e:.3e}")
    from tests._gates import gate
    gate(f"SDXL UNet {H * 8}x{W * 8} vs fp16-storage oracle", e, 5e-3)
    assert _rel(y[1], y[0]) > 1e-3


def test_unet_sdxl_forward_vs_oracle_1536(sdxl_model):
    """One whole UNet forward at 1536 x 1536 (192 x 192 latents: 9 216 / 2 304 tokens at the attention levels; the largest
    bucket of BASELINE configs[3]) vs the CPU oracle with fp16-storage emulation, rel-L2 <= 5e-3 (reference path
    src/models/unet.py:116-347).  ~31 TFLOP on the host cores."""
    _oracle_case(sdxl_model, 192, 192, 41)


@pytest.mark.skipif(os.environ.get("DS_TEST_2048") != "1", reason="five minutes of host cores: run with DS_TEST_2048=1 "
                    "(its log is committed as profiles/r05_unet_2048_vs_oracle.log)")
def test_unet_sdxl_forward_vs_oracle_2048(sdxl_model):
    """BASELINE configs[4]'s own shape: 2048 x 2048 (256 x 256 latents: 16 384 / 4 096 tokens), CFG batch 2, four character
    boxes.  ~72 TFLOP on the host cores - not part of the default suite."""
    _oracle_case(sdxl_model, 256, 256, 43)
