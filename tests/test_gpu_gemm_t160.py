"""GPU: gemm_t160_kernel (csrc/gemm_t160.hip) - 64 x 160 tiles, one block per CU, for the projections of a batch-1 request
(UNet batch 2 at 1024 x 1024: M = 2048, N = 1280, K = 1280 | 5120; the nn.Linear layers of diffusers' BasicTransformerBlock [3P]
reached from reference src/models/unet.py:244-338, attention_processor.py:84,209,261) - against a plain PyTorch fp32
reference of the same op and, bit for bit, against the 64 x 128 ring kernel it replaces at those shapes.

Tolerance vs fp32: max |err| <= 2e-3 max|ref| (one f16 rounding of an fp32-accumulated sum, as for every GEMM here).
"""
import ctypes as C
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


def _forced(lib, variant, fn):
    assert lib.ds_set_option(b"gemm_variant", variant) == 0
    try:
        return fn()
    finally:
        lib.ds_set_option(b"gemm_variant", 0)


def _gemm_op(lib, x, w, bias=None, residual=None, ln_partial=None, ln_c=None, ln_nstrips=0, stats_strip=0):
    """One DS_OP_GEMM through ds_op_run - the form the launch plan uses (i[10] = strips a consumer sums, i[11] = strip width a
    producer is asked for).  Returns (y, partial statistics or None, kernel name ds_op_describe reports)."""
    from diffsensei_amd.engine import make_op
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty((M, N), dtype=torch.float16, device=x.device)
    nent = 3 * (N // 160) if stats_strip == 160 else (N // stats_strip if stats_strip else 0)
    part = torch.zeros((nent, M, 2), dtype=torch.float32, device=x.device) if stats_strip else None
    op = make_op("GEMM", i=(M, N, K, K, 0, 1, 0, 1, 0, int(ln_partial is not None), ln_nstrips, stats_strip), f=(1e-5,),
                 l=(K, 0, K, N, N), p=(x, None, w, y, bias, None, residual, ln_partial, ln_c, part))
    name = C.create_string_buffer(128)
    fl, by = C.c_double(), C.c_double()
    assert lib.ds_op_describe(C.byref(op), name, 128, C.byref(fl), C.byref(by)) == 0
    rc = lib.ds_op_run(C.byref(op), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, lib.ds_last_error().decode()
    torch.cuda.synchronize()
    return y, part, name.value.decode()


@pytest.mark.parametrize("M,N,K,bias,res", [(2048, 1280, 1280, True, True), (2048, 1280, 5120, True, True), (2048, 1280, 1280, False, False),
                                            (1000, 320, 192, True, True), (8, 160, 64, True, False), (77, 480, 448, False, True)])
def test_t160_vs_fp32_and_vs_ring_kernel(hip_lib, M, N, K, bias, res):
    """Forced onto every shape it can run (whole and ragged row tiles, one k-tile to eighty, fewer k-tiles than ring stages):
    vs fp32 torch, and bit-identical to the automatic choice of round 5 (gemm_t160 off) - same MFMA, same k order, same epilogue
    arithmetic."""
    from diffsensei_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(M * 7 + N + K)
    x, w = _r((M, K), g), _r((N, K), g, 1 / math.sqrt(K))
    b = _r((N,), g) if bias else None
    r = (_r((M, N), g) * 2 + 0.5).half() if res else None
    dv = lambda t: None if t is None else t.to(DEV)
    ref = F.linear(x.float(), w.float(), None if b is None else b.float())
    if r is not None:
        ref = ref.half().float() + r.float()
    got = _forced(lib, 11, lambda: ops.gemm(dv(x), dv(w), dv(b), dv(r)))
    e = _relmax(got, ref)
    print(f"gemm_t160 M={M} N={N} K={K}: max err / max|ref| {e:.2e}")
    assert e <= 2e-3, e
    assert lib.ds_set_option(b"gemm_t160", 1) == 0
    try:
        old = ops.gemm(dv(x), dv(w), dv(b), dv(r))
    finally:
        lib.ds_set_option(b"gemm_t160", 0)
    assert torch.equal(got, old), "gemm_t160_kernel and the kernel it replaces differ"


def test_t160_is_the_automatic_choice_at_the_batch1_shapes_only(hip_lib):
    """The dispatch rule (host logic): one 64 x 160 block per CU where the 64 x 128 grid leaves CUs with two blocks."""
    from diffsensei_amd import _lib
    lib = _lib.load()
    fits = lambda m, n, k, b=1: int(lib.ds_gemm_t160_fits(m, n, k, b))
    assert fits(2048, 1280, 1280) and fits(2048, 1280, 5120)           # UNet batch 2 at 1024 x 1024, level 2
    assert fits(2048, 2560, 1280)                                       # q|k: 512 blocks of 64 x 160 -> 256 blocks of 128 x 160
    assert not fits(65536, 1280, 1280) and not fits(8192, 1280, 640)    # the benched batch; level-1 q|k of a batch-1 request (512 tall blocks)
    assert not fits(1024, 1280, 1280)                                   # 512 x 512: the 64 x 128 grid is already <= 256 blocks
    assert not fits(2048, 1280, 1280, 2) and not fits(2048, 1264, 1280)
    g = torch.Generator().manual_seed(1)
    x, w = _r((2048, 1280), g).to(DEV), _r((1280, 1280), g, 0.03).to(DEV)
    assert _gemm_op(lib, x, w)[2] == "gemm_t160_kernel"
    assert lib.ds_set_option(b"gemm_t160", 1) == 0
    try:
        assert not fits(2048, 1280, 1280) and _gemm_op(lib, x, w)[2] == "gemm_glds_kernel<64,false,3>"
    finally:
        lib.ds_set_option(b"gemm_t160", 0)


@pytest.mark.parametrize("M,K", [(2048, 1280), (2048, 5120), (1999, 1280)])
def test_t160_producer_statistics_and_consumer_chain(hip_lib, M, K):
    """The launch plan's sequence at UNet batch 2: out-projection + residual emitting gemm_t160_kernel's statistics format
    (i[11] = 160: entries of 64 | 64 | 32 columns per tile) -> attn2.to_q consuming the 24 entries per row (i[10] = 24), on
    gemm_t160_kernel both; and the same partials consumed by the 128-wide kernels (q|k / GEGLU consumers at this batch).  vs fp32 LayerNorm + linear; the stored producer output must be
    bit-identical to the GEMM without statistics."""
    from diffsensei_amd import _lib, ops
    from diffsensei_amd.engine import pack_ln_fused
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + K)
    Cc = 1280
    a, wo, bo = _r((M, K), g), _r((Cc, K), g, 1 / math.sqrt(K)), _r((Cc,), g)
    h0 = (_r((M, Cc), g) * 2 + 0.5).half()
    wq, gamma, beta = _r((Cc, Cc), g, 1 / math.sqrt(Cc)), (1 + 0.2 * torch.randn(Cc, generator=g)).half(), _r((Cc,), g, 0.2)
    dv = lambda t: t.to(DEV)
    h, part, name = _forced(lib, 11 if M % 64 else 0, lambda: _gemm_op(lib, dv(a), dv(wo), dv(bo), dv(h0), stats_strip=160))
    assert name == "gemm_t160_kernel" or M % 64
    plain = _forced(lib, 11, lambda: ops.gemm(dv(a), dv(wo), dv(bo), dv(h0)))
    assert torch.equal(h, plain), "statistics emission changed the stored output"
    hf = h.float().cpu()
    assert part.shape == (3 * (Cc // 160), M, 2)
    tiles = hf.view(M, Cc // 160, 160)
    ent = torch.stack([tiles[..., :64], tiles[..., 64:128]], 2)                       # [M, tiles, 2, 64]
    want_s = torch.cat([ent.sum(-1), tiles[..., 128:].sum(-1, keepdim=True)], 2).reshape(M, -1)
    want_q = torch.cat([(ent * ent).sum(-1), (tiles[..., 128:] ** 2).sum(-1, keepdim=True)], 2).reshape(M, -1)
    assert torch.allclose(part[..., 0].t().cpu(), want_s, rtol=1e-5, atol=1e-3)
    assert torch.allclose(part[..., 1].t().cpu(), want_q, rtol=1e-5, atol=1e-3)
    # consumer: LayerNorm(h) @ wq^T on the raw h
    gw, c2, b2 = pack_ln_fused(dv(wq), None, dv(gamma), dv(beta))
    ref = F.linear(F.layer_norm(hf, (Cc,), gamma.float(), beta.float(), 1e-5), wq.float())
    q_t160, _, nm = _forced(lib, 11, lambda: _gemm_op(lib, h, gw, b2, None, ln_partial=part, ln_c=c2, ln_nstrips=3 * (Cc // 160)))
    assert nm == "gemm_t160_kernel"
    assert lib.ds_set_option(b"gemm_t160", 1) == 0
    try:
        q_wide, _, nm2 = _gemm_op(lib, h, gw, b2, None, ln_partial=part, ln_c=c2, ln_nstrips=3 * (Cc // 160))
    finally:
        lib.ds_set_option(b"gemm_t160", 0)
    assert nm2.startswith("gemm_glds_kernel")
    e1, e2 = _relmax(q_t160, ref), _relmax(q_wide, ref)
    print(f"fused LayerNorm chain M={M} K={K}: consumer on gemm_t160 {e1:.2e}, on {nm2} {e2:.2e}")
    assert e1 <= 3e-3 and e2 <= 3e-3, (e1, e2)
    assert torch.equal(q_t160, q_wide), "the two consumer families sum the same partials to different bits"


def test_t160_refuses_what_it_does_not_implement(hip_lib):
    """64-column statistics from a direct caller (ds_gemm_ln_f16) never reach this kernel: the dispatch keeps the ring kernel."""
    from diffsensei_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    x, w, b = _r((2048, 1280), g).to(DEV), _r((1280, 1280), g, 0.03).to(DEV), _r((1280,), g).to(DEV)
    y, part = ops.gemm_ln(x, w, b, emit_stats=True)
    assert part.shape == (1280 // 64, 2048, 2)
    yf = y.float().cpu().view(2048, 20, 64)
    assert torch.allclose(part[..., 0].t().cpu(), yf.sum(-1), rtol=1e-5, atol=1e-3)
    assert torch.equal(y, ops.gemm(x, w, b))


@pytest.mark.parametrize("M,N,K,bias,res", [(2048, 2560, 1280, True, False), (2048, 1280, 1280, True, True), (1000, 320, 192, True, True),
                                            (130, 160, 64, False, False), (4096, 1280, 5120, True, True)])
def test_t160_tall_tiles_vs_fp32_and_vs_the_64_row_tiles(hip_lib, M, N, K, bias, res):
    """The 128 x 160 instantiation (four ring stages; the q|k projection of a batch-1 request: M = 2048, N = 2560 -> 16 x 16 = 256
    blocks), forced wherever the kernel runs: vs fp32 torch and bit-identical to the 64 x 160 tiles - same MFMA, same k order,
    same epilogue arithmetic; whole and ragged row tiles."""
    from diffsensei_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(M * 5 + N + K)
    x, w = _r((M, K), g), _r((N, K), g, 1 / math.sqrt(K))
    b = _r((N,), g) if bias else None
    r = (_r((M, N), g) * 2 + 0.5).half() if res else None
    dv = lambda t: None if t is None else t.to(DEV)
    ref = F.linear(x.float(), w.float(), None if b is None else b.float())
    if r is not None:
        ref = ref.half().float() + r.float()
    assert lib.ds_set_option(b"gemm_t160", 3) == 0
    try:
        got = _forced(lib, 11, lambda: ops.gemm(dv(x), dv(w), dv(b), dv(r)))
    finally:
        lib.ds_set_option(b"gemm_t160", 0)
    e = _relmax(got, ref)
    print(f"gemm_t160 (128-row tiles) M={M} N={N} K={K}: max err / max|ref| {e:.2e}")
    assert e <= 2e-3, e
    assert lib.ds_set_option(b"gemm_t160", 2) == 0
    try:
        short = _forced(lib, 11, lambda: ops.gemm(dv(x), dv(w), dv(b), dv(r)))
    finally:
        lib.ds_set_option(b"gemm_t160", 0)
    assert torch.equal(got, short), "the 128-row and the 64-row tiles of gemm_t160_kernel differ"


def test_t160_tall_tiles_rule_and_fused_layernorm_chain(hip_lib):
    """The rule picks 128-row tiles for the q|k projection of a batch-1 request only where the 64-row grid overflows the CUs;
    producer (statistics, 160-column format) and consumer (partial sums) forms on 128-row tiles give the bits of the 64-row tiles."""
    from diffsensei_amd import _lib
    from diffsensei_amd.engine import pack_ln_fused
    lib = _lib.load()
    fits = lambda m, n, k, b=1: int(lib.ds_gemm_t160_fits(m, n, k, b))
    assert fits(2048, 2560, 1280) == 1 and fits(2048, 1280, 1280) == 1 and fits(65536, 2560, 1280) == 0
    g = torch.Generator().manual_seed(77)
    M, Cc = 2048, 1280
    a, wo, bo = _r((M, Cc), g).to(DEV), _r((Cc, Cc), g, 1 / math.sqrt(Cc)).to(DEV), _r((Cc,), g).to(DEV)
    h0 = (_r((M, Cc), g) * 2 + 0.5).half().to(DEV)
    wq, gamma, beta = _r((2 * Cc, Cc), g, 1 / math.sqrt(Cc)).to(DEV), (1 + 0.2 * torch.randn(Cc, generator=g)).half().to(DEV), _r((Cc,), g, 0.2).to(DEV)
    gw, c2, b2 = pack_ln_fused(wq, None, gamma, beta)
    outs = {}
    for mode in (2, 3):
        assert lib.ds_set_option(b"gemm_t160", mode) == 0 and lib.ds_set_option(b"gemm_variant", 11) == 0
        try:
            h, part, nm = _gemm_op(lib, a, wo, bias=bo, residual=h0, stats_strip=160)
            y, _, nm2 = _gemm_op(lib, h, gw, bias=b2, ln_partial=part, ln_c=c2, ln_nstrips=24)
        finally:
            lib.ds_set_option(b"gemm_t160", 0)
            lib.ds_set_option(b"gemm_variant", 0)
        assert nm.startswith("gemm_t160_kernel") and nm2.startswith("gemm_t160_kernel"), (nm, nm2)
        outs[mode] = (h, part, y)
    for u, v in zip(outs[2], outs[3]):
        assert torch.equal(u, v)
    hr = ((a.float() @ wo.float().t() + bo.float()).half().float() + h0.float()).half().float()
    ref = F.linear(F.layer_norm(hr, (Cc,), gamma.float(), beta.float(), 1e-5), wq.float())
    assert _relmax(outs[3][2], ref) <= 3e-3
    # automatic dispatch at the q|k shape of a batch-1 request
    assert _gemm_op(lib, outs[3][0], gw, bias=b2, ln_partial=outs[3][1], ln_c=c2, ln_nstrips=24)[2] == "gemm_t160_kernel<128 rows>"
