"""GPU: LayerNorm folded into the GEMM pair around it (csrc/gemm_pp.hip "LayerNorm", include/diffsensei_hip.h ds_gemm_ln_f16)
vs plain PyTorch fp32 references of the ops it replaces - nn.LayerNorm followed by nn.Linear / GEGLU in diffusers'
BasicTransformerBlock [3P], reached from reference src/models/unet.py:244-338 (norm2 -> attn2.to_q, norm3 -> ff.net.0).

Tolerances: row statistics (fp32 sums of f16 values) - mean 1e-5 absolute + 1e-5 relative, rstd 1e-4 relative; fused
consumer output vs fp32 LayerNorm + linear on the same f16 inputs: max |err| <= 3e-3 max|ref| (the unfused HIP pair
LayerNorm kernel -> GEMM is measured beside it: same size of error - one path rounds LN(x) to f16, the other gamma (.) W);
rows with a mean of 8 standard deviations still inside 5e-3.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _r(shape, g, scale=1.0):
    return (torch.randn(shape, generator=g) * scale).half()


def _relmax(got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    assert got.shape == ref.shape and torch.isfinite(got).all()
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-6)).item()


@pytest.mark.parametrize("M,N,K,res", [(512, 1280, 1280, True), (1024, 256, 128, False), (2048, 1280, 5120, True),
                                       (1024, 640, 640, True), (768, 320, 256, False), (32768, 640, 640, True)])
def test_producer_row_statistics(hip_lib, M, N, K, res):
    """The +residual projection that writes the residual stream also emits, per row and 64-column strip, the (sum, sum of
    squares) of the f16 values it stores; `ln_finalize` turns them into (mean, rstd).  The stored output itself must be
    bit-identical to the GEMM without statistics."""
    from diffsensei_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    x, w, b = _r((M, K), g), _r((N, K), g, 1 / math.sqrt(K)), _r((N,), g)
    r = (_r((M, N), g) * 3 + 1.5).half() if res else None
    dv = lambda t: None if t is None else t.to(DEV)
    lib = __import__("diffsensei_amd._lib", fromlist=["x"]).load()
    lib.ds_set_option(b"gemm_variant", 3)   # (N % 256 != 0: whole 64-column strips of a ragged last tile column, round 6)
    try:
        y, part = ops.gemm_ln(dv(x), dv(w), dv(b), residual=dv(r), emit_stats=True)
        assert part.shape == (N // 64, M, 2)
        y0 = ops.gemm(dv(x), dv(w), dv(b), dv(r))
    finally:
        lib.ds_set_option(b"gemm_variant", 0)
    assert torch.equal(y, y0), "statistics emission changed the stored output"
    yf = y.float().cpu()
    strips = yf.view(M, N // 64, 64)
    assert torch.allclose(part[..., 0].t().cpu(), strips.sum(-1), rtol=1e-5, atol=1e-3)
    assert torch.allclose(part[..., 1].t().cpu(), (strips * strips).sum(-1), rtol=1e-5, atol=1e-3)
    st = ops.ln_finalize(part, N, 1e-5).cpu()
    mean, var = yf.mean(1), yf.var(1, unbiased=False)
    assert torch.allclose(st[:, 0], mean, rtol=1e-5, atol=1e-5)
    assert torch.allclose(st[:, 1], torch.rsqrt(var + 1e-5), rtol=1e-4, atol=0)


def _pack(w, bias, gamma, beta):
    from diffsensei_amd.engine import pack_ln_fused
    return pack_ln_fused(w.to(DEV), None if bias is None else bias.to(DEV), gamma.to(DEV), beta.to(DEV))


@pytest.mark.parametrize("M,N,K,offset", [(512, 1280, 1280, 0.0), (256, 256, 128, 0.0), (1024, 2560, 1280, 0.3), (512, 1280, 1280, 8.0),
                                          (1024, 640, 640, 0.3), (24576, 640, 640, 0.0), (512, 192, 128, 0.0)])
def test_consumer_plain_vs_layernorm_linear(hip_lib, M, N, K, offset):
    """y = rstd (x gw^T - mean c) + b' on the raw x  ==  Linear(LayerNorm(x)); `offset`: row mean in standard deviations
    (the rank-1 term is carried as an f16 (hi, lo) pair, so a large mean must not cost accuracy)."""
    from diffsensei_amd import ops
    g = torch.Generator().manual_seed(M + N + K + int(offset * 10))
    x = ((torch.randn((M, K), generator=g) + offset) * (1.0 + torch.rand((M, 1), generator=g) * 3)).half()
    w, gamma, beta = _r((N, K), g, 1 / math.sqrt(K)), (1 + 0.2 * torch.randn(K, generator=g)).half(), _r((K,), g, 0.2)
    ref = F.linear(F.layer_norm(x.float(), (K,), gamma.float(), beta.float(), 1e-5), w.float())
    gw, c2, b2 = _pack(w, None, gamma, beta)
    xd = x.to(DEV)
    # statistics from a producer GEMM would be those of its output; here: of x itself, through the same finalize kernel
    xs = x.float().view(M, K // 64, 64)
    part = torch.stack([xs.sum(-1).t(), (xs * xs).sum(-1).t()], dim=-1).contiguous().to(DEV)
    st = ops.ln_finalize(part, K, 1e-5)
    got = ops.gemm_ln(xd, gw, b2, c2, st)
    unfused = ops.gemm(ops.layernorm(xd, gamma.to(DEV), beta.to(DEV), 1e-5), w.to(DEV))
    e_f, e_u = _relmax(got, ref), _relmax(unfused, ref)
    print(f"LN -> linear M={M} N={N} K={K} mean/std {offset}: fused {e_f:.2e}, LayerNorm kernel + GEMM {e_u:.2e}")
    assert e_f <= (3e-3 if offset < 1 else 5e-3), e_f


def test_consumer_geglu_vs_layernorm_geglu(hip_lib):
    """norm3 -> ff.net.0 (GEGLU, packed weights): h * gelu(g) of Linear(LayerNorm(x)), fused, vs fp32."""
    from diffsensei_amd import ops
    from diffsensei_amd.engine import pack_geglu, pack_ln_fused
    g = torch.Generator().manual_seed(5)
    M, C = 512, 256
    x = ((torch.randn((M, C), generator=g) + 0.2) * 2).half()
    w, b = _r((8 * C, C), g, 1 / math.sqrt(C)), _r((8 * C,), g, 0.3)
    gamma, beta = (1 + 0.2 * torch.randn(C, generator=g)).half(), _r((C,), g, 0.2)
    z = F.linear(F.layer_norm(x.float(), (C,), gamma.float(), beta.float(), 1e-5), w.float(), b.float())
    ref = z[:, :4 * C].half().float() * F.gelu(z[:, 4 * C:].half().float())
    gw, c2, b2 = pack_ln_fused(w.to(DEV), b.to(DEV), gamma.to(DEV), beta.to(DEV))
    gwp, b2p = pack_geglu(gw, b2)
    half = 4 * C
    c2p = torch.stack([c2[:half].reshape(-1, 64, 2), c2[half:].reshape(-1, 64, 2)], dim=1).reshape(-1, 2).contiguous()
    xs = x.float().view(M, C // 64, 64)
    part = torch.stack([xs.sum(-1).t(), (xs * xs).sum(-1).t()], dim=-1).contiguous().to(DEV)
    st = ops.ln_finalize(part, C, 1e-5)
    got = ops.gemm_ln(x.to(DEV), gwp, b2p, c2p, st, geglu=True)
    e = _relmax(got, ref)
    print(f"LN -> GEGLU fused: {e:.2e}")
    assert got.shape == (M, 4 * C) and e <= 4e-3, e


def test_fused_chain_producer_finalize_consumer_and_refusals(hip_lib):
    """The sequence the launch plan emits: out-projection + residual (statistics) -> finalize -> fused to_q, against the fp32
    chain; 30 back-to-back repetitions must give the same bits (the pieces ride the tile hand-over's counted waits); shapes or
    options the fused epilogues do not implement are refused, not silently mis-served."""
    from diffsensei_amd import _lib, ops
    g = torch.Generator().manual_seed(11)
    M, C = 2048, 1280
    a, wo, bo = _r((M, C), g), _r((C, C), g, 1 / math.sqrt(C)), _r((C,), g)
    h0 = (_r((M, C), g) * 2 + 0.5).half()
    wq, gamma, beta = _r((C, C), g, 1 / math.sqrt(C)), (1 + 0.2 * torch.randn(C, generator=g)).half(), _r((C,), g, 0.2)
    dv = lambda t: t.to(DEV)
    h, part = ops.gemm_ln(dv(a), dv(wo), dv(bo), residual=dv(h0), emit_stats=True)
    st = ops.ln_finalize(part, C, 1e-5)
    gw, c2, b2 = _pack(wq, None, gamma, beta)
    q = ops.gemm_ln(h, gw, b2, c2, st)
    hr = (a.float() @ wo.float().t() + bo.float()).half().float() + h0.float()
    ref = F.linear(F.layer_norm(hr.half().float(), (C,), gamma.float(), beta.float(), 1e-5), wq.float())
    e = _relmax(q, ref)
    print(f"producer -> finalize -> consumer chain: {e:.2e}")
    assert e <= 3e-3, e
    for _ in range(30):
        h2, part2 = ops.gemm_ln(dv(a), dv(wo), dv(bo), residual=dv(h0), emit_stats=True)
        q2 = ops.gemm_ln(h2, gw, b2, c2, ops.ln_finalize(part2, C, 1e-5))
        assert torch.equal(h2, h) and torch.equal(part2, part) and torch.equal(q2, q)
    # what the dispatch gives a shape: 1 = the 256 x 256 kernel, 2 = the 128-wide kernels (small batches, 640 channels), 0 = none
    assert ops.gemm_ln_fusable(65536, 1280, 1280) == 1 and ops.gemm_ln_fusable(65536, 10240, 1280, geglu=True) == 1
    assert ops.gemm_ln_fusable(2048, 1280, 1280) == 2 and ops.gemm_ln_fusable(8192, 640, 640) == 2
    assert ops.gemm_ln_fusable(65536, 640, 640) == 1    # round 6: whole 64-column strips of the ragged third tile column
    assert ops.gemm_ln_fusable(2048, 1280, 1288) == 0                # K not a multiple of 64: the register-staged fallback
    with pytest.raises(_lib.DiffSenseiHipError):          # a producer needs whole 64-column strips per 128-column tile
        ops.gemm_ln(dv(a), dv(wo[:1160]), dv(bo[:1160]), residual=dv(h0[:, :1160].contiguous()), emit_stats=True)
    with pytest.raises(_lib.DiffSenseiHipError):          # ragged M: the generic epilogue has no fused form
        ops.gemm_ln(dv(a[:2000]), gw, b2, c2, st[:2000].contiguous())
    with pytest.raises(_lib.DiffSenseiHipError):          # consumer without its b'
        ops.gemm_ln(dv(a), gw, None, c2, st)


@pytest.mark.parametrize("Z,N,C", [(3, 512, 256), (3, 512, 640), (2, 1024, 384), (5, 256, 1280)])
def test_consumer_swapped_vs_layernorm_linear(hip_lib, Z, N, C):
    """norm1 -> attn1.to_v, produced transposed per image: out[z] = Wv LN(x[z])^T (operand-swapped form: the statistics run
    along the output columns, c and b' along its rows, batch items folded into the persistent tile walk).  C = 640 / 384: the
    output channels are the tile ROWS, 2.5 / 1.5 tiles - 32-row pieces past C are skipped (round 6); the rows behind the last
    image's C rows must stay untouched, and repeated launches give the same bits (the skipped pieces' stores are padded in the
    counted waits of the tile hand-over)."""
    from diffsensei_amd import ops
    from diffsensei_amd.engine import pack_ln_fused
    g = torch.Generator().manual_seed(21 + C)
    x = ((torch.randn((Z, N, C), generator=g) + 0.4) * (1.0 + torch.rand((Z, N, 1), generator=g) * 3)).half()
    wv, gamma, beta = _r((C, C), g, 1 / math.sqrt(C)), (1 + 0.2 * torch.randn(C, generator=g)).half(), _r((C,), g, 0.3)
    ref = torch.einsum("ck,znk->zcn", wv.float(), F.layer_norm(x.float(), (C,), gamma.float(), beta.float(), 1e-5))
    gw, c2, _ = pack_ln_fused(wv.to(DEV), None, gamma.to(DEV), beta.to(DEV))
    bf = (wv.double() @ beta.double()).float()
    bh = bf.half()
    cb = torch.cat([c2.cpu(), torch.stack([bh, (bf - bh.float()).half()], dim=1)], dim=1).contiguous().to(DEV)
    xs = x.float().view(Z * N, C // 64, 64)
    part = torch.stack([xs.sum(-1).t(), (xs * xs).sum(-1).t()], dim=-1).contiguous().to(DEV)
    st = ops.ln_finalize(part, C, 1e-5)
    buf = torch.full((Z * C * N + 4096,), 3.0, dtype=torch.float16, device=DEV)     # canary behind the last image's rows
    got = ops.gemm_ln_swapped(gw, x.to(DEV), st, cb, out=buf[:Z * C * N].view(Z, C, N))
    assert bool((buf[Z * C * N:] == 3.0).all())
    for _ in range(5):
        assert torch.equal(ops.gemm_ln_swapped(gw, x.to(DEV), st, cb), got)
    tn = ops.layernorm(x.view(Z * N, C).to(DEV), gamma.to(DEV), beta.to(DEV), 1e-5).view(Z, N, C)
    unfused = ops.gemm_batched_nt(wv.to(DEV), tn)
    e_f, e_u = _relmax(got, ref), _relmax(unfused, ref)
    print(f"LN -> V^T (swapped) Z={Z} N={N} C={C}: fused {e_f:.2e}, LayerNorm kernel + GEMM {e_u:.2e}")
    assert got.shape == (Z, C, N) and e_f <= 3e-3, e_f


# ---- the same pair on the 128-wide LDS-DMA kernels (csrc/gemm.hip "Fused LayerNorm"): small batches and the 640-channel level
@pytest.mark.parametrize("M,N,K,res", [(2048, 1280, 1280, True), (8192, 640, 640, True), (2048, 1280, 5120, True),
                                       (1000, 640, 2560, False), (154, 1280, 1280, True)])
def test_wide_producer_row_statistics(hip_lib, M, N, K, res):
    """Producer on the 64 x 128 / 128 x 128 kernels (ring-buffered and one-buffer variants, ragged M): same statistics format
    as gemm_pp_kernel's, the stored output bit-identical to the GEMM without statistics."""
    from diffsensei_amd import ops
    assert ops.gemm_ln_fusable(M, N, K) == 2
    g = torch.Generator().manual_seed(M + N + K + 1)
    x, w, b = _r((M, K), g), _r((N, K), g, 1 / math.sqrt(K)), _r((N,), g)
    r = (_r((M, N), g) * 3 + 1.5).half() if res else None
    dv = lambda t: None if t is None else t.to(DEV)
    y, part = ops.gemm_ln(dv(x), dv(w), dv(b), residual=dv(r), emit_stats=True)
    assert torch.equal(y, ops.gemm(dv(x), dv(w), dv(b), dv(r))), "statistics emission changed the stored output"
    yf = y.float().cpu()
    strips = yf.view(M, N // 64, 64)
    assert torch.allclose(part[..., 0].t().cpu(), strips.sum(-1), rtol=1e-5, atol=1e-3)
    assert torch.allclose(part[..., 1].t().cpu(), (strips * strips).sum(-1), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("M,N,K,offset", [(2048, 1280, 1280, 0.0), (8192, 640, 640, 0.3), (1000, 1280, 640, 0.0),
                                          (2048, 2560, 1280, 8.0), (64, 128, 64, 0.0)])
def test_wide_consumer_plain_vs_layernorm_linear(hip_lib, M, N, K, offset):
    """Consumer that sums the partials of its own rows (no finalize launch) vs fp32 LayerNorm + linear, and vs the 256 x 256
    consumer fed by the finalize launch where that one takes the shape: the two agree to the last bits of an f16 almost
    everywhere (same statistics bits by construction; the rank-1 term is an fp32 fma here, an f16 (hi, lo) MFMA there)."""
    from diffsensei_amd import ops
    g = torch.Generator().manual_seed(M + N + K + int(offset * 10) + 3)
    x = ((torch.randn((M, K), generator=g) + offset) * (1.0 + torch.rand((M, 1), generator=g) * 3)).half()
    w, gamma, beta = _r((N, K), g, 1 / math.sqrt(K)), (1 + 0.2 * torch.randn(K, generator=g)).half(), _r((K,), g, 0.2)
    b = _r((N,), g, 0.3)
    ref = F.linear(F.layer_norm(x.float(), (K,), gamma.float(), beta.float(), 1e-5), w.float(), b.float())
    gw, c2, b2 = _pack(w, b, gamma, beta)
    xs = x.float().view(M, K // 64, 64)
    part = torch.stack([xs.sum(-1).t(), (xs * xs).sum(-1).t()], dim=-1).contiguous().to(DEV)
    got = ops.gemm_ln_partial(x.to(DEV), gw, b2, c2, part)
    e = _relmax(got, ref)
    line = f"LN -> linear (128-wide kernels) M={M} N={N} K={K} mean/std {offset}: {e:.2e}"
    if M % 256 == 0 and N % 256 == 0 and K % 128 == 0:
        pp = ops.gemm_ln(x.to(DEV), gw, b2, c2, ops.ln_finalize(part, K, 1e-5))
        d = (got.float() - pp.float()).abs().max().item()
        line += f"; vs the 256 x 256 consumer: max |diff| {d:.2e}, {(got != pp).float().mean().item():.2e} of the values differ"
        assert d <= 2e-3 * ref.abs().max().item()
    print(line)
    assert e <= (3e-3 if offset < 1 else 5e-3), e


def test_wide_consumer_geglu_and_chain(hip_lib):
    """norm3 -> ff.net.0 (GEGLU, packed weights) on the 128-wide kernels, fed by a producer of the same family: the sequence a
    small-batch launch plan emits (out-projection + residual with statistics -> GEGLU consumer, NO launch in between), vs fp32;
    20 repetitions give the same bits; a producer of one family feeds a consumer of the other."""
    from diffsensei_amd import ops
    from diffsensei_amd.engine import pack_geglu, pack_ln_fused
    g = torch.Generator().manual_seed(23)
    M, C = 2048, 640
    a, wo, bo = _r((M, C), g), _r((C, C), g, 1 / math.sqrt(C)), _r((C,), g)
    h0 = (_r((M, C), g) * 2 + 0.5).half()
    w, b = _r((8 * C, C), g, 1 / math.sqrt(C)), _r((8 * C,), g, 0.3)
    gamma, beta = (1 + 0.2 * torch.randn(C, generator=g)).half(), _r((C,), g, 0.2)
    dv = lambda t: t.to(DEV)
    h, part = ops.gemm_ln(dv(a), dv(wo), dv(bo), residual=dv(h0), emit_stats=True)
    gw, c2, b2 = pack_ln_fused(dv(w), dv(b), dv(gamma), dv(beta))
    gwp, b2p = pack_geglu(gw, b2)
    half = 4 * C
    c2p = torch.stack([c2[:half].reshape(-1, 64, 2), c2[half:].reshape(-1, 64, 2)], dim=1).reshape(-1, 2).contiguous()
    got = ops.gemm_ln_partial(h, gwp, b2p, c2p, part, geglu=True)
    hr = ((a.float() @ wo.float().t() + bo.float()).half().float() + h0.float()).half().float()
    z = F.linear(F.layer_norm(hr, (C,), gamma.float(), beta.float(), 1e-5), w.float(), b.float())
    ref = z[:, :4 * C].half().float() * F.gelu(z[:, 4 * C:].half().float())
    e = _relmax(got, ref)
    print(f"128-wide producer -> GEGLU consumer: {e:.2e}")
    assert got.shape == (M, 4 * C) and e <= 4e-3, e
    for _ in range(20):
        h2, part2 = ops.gemm_ln(dv(a), dv(wo), dv(bo), residual=dv(h0), emit_stats=True)
        assert torch.equal(h2, h) and torch.equal(part2, part)
        assert torch.equal(ops.gemm_ln_partial(h2, gwp, b2p, c2p, part2, geglu=True), got)
    # mixed families: 128-wide producer -> finalize -> 256 x 256 consumer (the 640-channel level at UNet batch 64: N = 5120)
    mixed = ops.gemm_ln(h, gwp, b2p, c2p, ops.ln_finalize(part, C, 1e-5), geglu=True)
    assert _relmax(mixed, ref) <= 4e-3
    with pytest.raises(Exception):                         # partial sums of the wrong width
        ops.gemm_ln_partial(h, gwp, b2p, c2p, part[:5].contiguous(), geglu=True)



@pytest.mark.parametrize("Z,N,C", [(2, 1024, 1280), (3, 296, 640), (2, 4096, 640), (1, 72, 128)])
def test_wide_consumer_swapped_vs_layernorm_linear(hip_lib, Z, N, C):
    """norm1 -> attn1.to_v on the 128-wide kernels (small batches; any token count, e.g. 296):
    every block finalises the statistics of its 128 output columns from the producer's partial sums.  vs fp32, and vs the
    256 x 256 form where that one takes the shape."""
    from diffsensei_amd import ops
    from diffsensei_amd.engine import pack_ln_fused
    g = torch.Generator().manual_seed(Z * 7 + N + C)
    x = ((torch.randn((Z, N, C), generator=g) + 0.4) * (1.0 + torch.rand((Z, N, 1), generator=g) * 3)).half()
    wv, gamma, beta = _r((C, C), g, 1 / math.sqrt(C)), (1 + 0.2 * torch.randn(C, generator=g)).half(), _r((C,), g, 0.3)
    ref = torch.einsum("ck,znk->zcn", wv.float(), F.layer_norm(x.float(), (C,), gamma.float(), beta.float(), 1e-5))
    gw, c2, _ = pack_ln_fused(wv.to(DEV), None, gamma.to(DEV), beta.to(DEV))
    bf = (wv.double() @ beta.double()).float()
    bh = bf.half()
    cb = torch.cat([c2.cpu(), torch.stack([bh, (bf - bh.float()).half()], dim=1)], dim=1).contiguous().to(DEV)
    xs = x.float().view(Z * N, C // 64, 64)
    part = torch.stack([xs.sum(-1).t(), (xs * xs).sum(-1).t()], dim=-1).contiguous().to(DEV)
    got = ops.gemm_ln_swapped_partial(gw, x.to(DEV), part, cb)
    e = _relmax(got, ref)
    line = f"LN -> V^T (swapped, 128-wide kernels) Z={Z} N={N} C={C}: {e:.2e}"
    if N % 256 == 0 and C % 256 == 0:
        pp = ops.gemm_ln_swapped(gw, x.to(DEV), ops.ln_finalize(part, C, 1e-5), cb)
        d = (got.float() - pp.float()).abs().max().item()
        line += f"; vs the 256 x 256 form: max |diff| {d:.2e}"
        assert d <= 2e-3 * ref.abs().max().item()
    print(line)
    assert got.shape == (Z, C, N) and e <= 3e-3, e
