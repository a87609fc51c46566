"""GPU: gemm_g320_kernel (csrc/gemm_g320.hip) - 256 x 320 tiles, one block per CU, the GEGLU feed-forward projection of a
batch-1 request (UNet batch 2 at 1024 x 1024: M = 2048, N = 10240 packed, K = 1280; diffusers' GEGLU [3P] inside
BasicTransformerBlock, reached from reference src/models/unet.py:244-338) - against a plain PyTorch fp32 reference of the same
op and, bit for bit, against the 128-row-packed GEGLU kernels it replaces at that shape.

Tolerance vs fp32: max |err| <= 3e-3 max|ref| plain, 4e-3 behind a fused LayerNorm (the tolerances of tests/test_gpu_ops.py /
tests/test_gpu_ln_fusion.py for the same epilogue).
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


@pytest.mark.parametrize("M,Ch,K", [(2048, 5120, 1280), (256, 160, 64), (300, 320, 192), (77, 480, 128), (2048, 2560, 640)])
def test_g320_geglu_vs_fp32_and_vs_the_128_packed_kernels(hip_lib, M, Ch, K):
    """h * gelu(g) of x W^T + b, W in 320-row groups: whole and ragged row tiles, one k-tile to twenty, one column tile to 32;
    vs fp32 torch, and - where the inner width also packs in 128-row groups - bit-identical to ops.gemm(geglu=True): same MFMA,
    same k order, same epilogue arithmetic (f16 h and g, f16 gelu(g), f16 product)."""
    from diffsensei_amd import ops
    from diffsensei_amd.engine import pack_geglu, pack_geglu320
    g = torch.Generator().manual_seed(M * 3 + Ch + K)
    x, w, b = _r((M, K), g), _r((2 * Ch, K), g, 1 / math.sqrt(K)), _r((2 * Ch,), g, 0.3)
    z = F.linear(x.float(), w.float(), b.float())
    ref = z[:, :Ch].half().float() * F.gelu(z[:, Ch:].half().float())
    dv = lambda t: t.to(DEV)
    got = ops.gemm(dv(x), pack_geglu320(dv(w)), pack_geglu320(dv(b)), geglu=320)
    e = _relmax(got, ref)
    print(f"gemm_g320 M={M} N={2 * Ch} K={K}: max err / max|ref| {e:.2e}")
    assert got.shape == (M, Ch) and e <= 3e-3, e
    if Ch % 64 == 0:
        wp, bp = pack_geglu(dv(w), dv(b))
        old = ops.gemm(dv(x), wp, bp, geglu=True)
        assert torch.equal(got, old), "gemm_g320_kernel and the 128-packed GEGLU kernels differ"
    nob = ops.gemm(dv(x), pack_geglu320(dv(w)), None, geglu=320)
    z0 = F.linear(x.float(), w.float())
    assert _relmax(nob, z0[:, :Ch].half().float() * F.gelu(z0[:, Ch:].half().float())) <= 3e-3


def test_g320_consumes_a_fused_layernorm_like_the_128_wide_kernels(hip_lib):
    """The launch plan's sequence at UNet batch 2: out-projection + residual emitting row statistics (64-column strips here;
    160-column strips of a gemm_t160_kernel producer below) -> GEGLU consumer on gemm_g320_kernel, no launch in between; vs fp32
    LayerNorm + Linear + GEGLU, bit-identical to the 128-packed consumer, 10 repetitions give the same bits."""
    from diffsensei_amd import _lib, ops
    from diffsensei_amd.engine import make_op, pack_geglu, pack_geglu320, pack_ln_fused
    lib = _lib.load()
    g = torch.Generator().manual_seed(41)
    M, Cc = 2048, 1280
    a, wo, bo = _r((M, Cc), g), _r((Cc, Cc), g, 1 / math.sqrt(Cc)), _r((Cc,), g)
    h0 = (_r((M, Cc), g) * 2 + 0.5).half()
    w, b = _r((8 * Cc, Cc), g, 1 / math.sqrt(Cc)), _r((8 * Cc,), g, 0.3)
    gamma, beta = (1 + 0.2 * torch.randn(Cc, generator=g)).half(), _r((Cc,), g, 0.2)
    dv = lambda t: t.to(DEV)
    h, part = ops.gemm_ln(dv(a), dv(wo), dv(bo), residual=dv(h0), emit_stats=True)
    gw, c2, b2 = pack_ln_fused(dv(w), dv(b), dv(gamma), dv(beta))
    got = ops.gemm_ln_partial(h, pack_geglu320(gw), pack_geglu320(b2), pack_geglu320(c2), part, geglu=320)
    hr = ((a.float() @ wo.float().t() + bo.float()).half().float() + h0.float()).half().float()
    z = F.linear(F.layer_norm(hr, (Cc,), gamma.float(), beta.float(), 1e-5), w.float(), b.float())
    ref = z[:, :4 * Cc].half().float() * F.gelu(z[:, 4 * Cc:].half().float())
    e = _relmax(got, ref)
    print(f"producer -> gemm_g320 GEGLU consumer: {e:.2e}")
    assert got.shape == (M, 4 * Cc) and e <= 4e-3, e
    gwp, b2p = pack_geglu(gw, b2)
    half = 4 * Cc
    c2p = torch.stack([c2[:half].reshape(-1, 64, 2), c2[half:].reshape(-1, 64, 2)], dim=1).reshape(-1, 2).contiguous()
    old = ops.gemm_ln_partial(h, gwp, b2p, c2p, part, geglu=True)
    assert torch.equal(got, old), "gemm_g320_kernel and the 128-packed consumer differ"
    for _ in range(10):
        assert torch.equal(ops.gemm_ln_partial(h, pack_geglu320(gw), pack_geglu320(b2), pack_geglu320(c2), part, geglu=320), got)

    # the plan's form: a gemm_t160_kernel producer (three entries per 160 columns: 24 per row) feeding it through DS_OP_GEMM
    def op_gemm(x, wt, y, N, K, epi, bias=None, residual=None, ln_partial=None, ln_c=None, nstr=0, strip=0, part_out=None):
        op = make_op("GEMM", i=(M, N, K, K, epi, 1, 0, 1, 0, int(ln_partial is not None), nstr, strip), f=(1e-5,),
                     l=(K, 0, K, y.shape[1], y.shape[1]), p=(x, None, wt, y, bias, None, residual, ln_partial, ln_c, part_out))
        name = C.create_string_buffer(128)
        fl, by = C.c_double(), C.c_double()
        assert lib.ds_op_describe(C.byref(op), name, 128, C.byref(fl), C.byref(by)) == 0
        rc = lib.ds_op_run(C.byref(op), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, lib.ds_last_error().decode()
        torch.cuda.synchronize()
        return name.value.decode()

    h160 = torch.empty((M, Cc), dtype=torch.float16, device=DEV)
    part160 = torch.zeros((24, M, 2), dtype=torch.float32, device=DEV)
    assert op_gemm(dv(a), dv(wo), h160, Cc, Cc, 0, bias=dv(bo), residual=dv(h0), strip=160, part_out=part160) == "gemm_t160_kernel"
    assert torch.equal(h160, h)
    y = torch.empty((M, 4 * Cc), dtype=torch.float16, device=DEV)
    nm = op_gemm(h160, pack_geglu320(gw), y, 8 * Cc, Cc, 4, bias=pack_geglu320(b2), ln_partial=part160, ln_c=pack_geglu320(c2), nstr=24)
    assert nm == "gemm_g320_kernel", nm
    e2 = _relmax(y, ref)
    print(f"gemm_t160 producer (160-column statistics) -> gemm_g320 consumer: {e2:.2e}")
    assert e2 <= 4e-3, e2


def test_g320_dispatch_rule_and_refusals(hip_lib):
    """Pure host logic + the launcher's refusals: the rule picks the shape of a batch-1 request only; the option switches it off;
    a residual, a packed width that is no multiple of 320 or finalised statistics are refused loudly."""
    from diffsensei_amd import _lib, ops
    from diffsensei_amd.engine import pack_geglu320
    lib = _lib.load()
    fits = lambda m, n, k, b=1: int(lib.ds_gemm_g320_fits(m, n, k, b))
    assert fits(2048, 10240, 1280) == 1                      # UNet batch 2, 1024 x 1024, 1280-channel level: 8 x 32 = 256 blocks
    assert fits(2048, 10240, 1280, 2) == 0 and fits(65536, 10240, 1280) == 0 and fits(8192, 5120, 640) == 0
    assert fits(512, 10240, 1280) == 0                       # 64 blocks: too few CUs busy
    assert fits(2048, 10240, 1288) == 0 and fits(2048, 10112, 1280) == 0
    assert lib.ds_set_option(b"gemm_g320", 1) == 0
    try:
        assert fits(2048, 10240, 1280) == 0
    finally:
        lib.ds_set_option(b"gemm_g320", 0)
    assert ops.gemm_ln_fusable(2048, 10240, 1280, geglu=320) == 2
    g = torch.Generator().manual_seed(3)
    x, w = _r((256, 64), g).to(DEV), _r((640, 64), g).to(DEV)
    with pytest.raises(Exception):
        ops.gemm(x, pack_geglu320(w), None, residual=torch.zeros((256, 320), dtype=torch.float16, device=DEV), geglu=320)
    with pytest.raises(Exception):
        ops.gemm(x, w[:384].contiguous(), None, geglu=320)


def test_g320_plain_form_for_the_one_block_per_cu_qk_projection(hip_lib):
    """The plain-epilogue instantiation: q|k at M = 8192, N = 2560, K = 1280 (UNet batch 8 at 1024 x 1024, batch 2 at 2048 x 2048:
    8 x ... 32 x 8 = 256 tiles of 256 x 320) is dispatched to it automatically - bias form and fused-LayerNorm consumer form vs fp32
    torch and bit-identical to the 128 x 128 kernel it replaces there (option gemm_g320 = 1 switches the rule off); other shapes
    keep their kernels."""
    from diffsensei_amd import _lib, ops
    from diffsensei_amd.engine import make_op, pack_ln_fused
    lib = _lib.load()
    g = torch.Generator().manual_seed(99)
    M, N, K = 8192, 2560, 1280
    x, w, b = _r((M, K), g).to(DEV), _r((N, K), g, 1 / math.sqrt(K)).to(DEV), _r((N,), g, 0.3).to(DEV)

    def name_of(m, n, k):
        y = torch.empty((8, 8), dtype=torch.float16, device=DEV)
        op = make_op("GEMM", i=(m, n, k, k, 0, 1, 0, 1, 0, 0, 0, 0), f=(1e-5,), l=(k, 0, k, n, n), p=(x, None, w, y, None, None, None, None, None, None))
        nm = C.create_string_buffer(128)
        fl, by = C.c_double(), C.c_double()
        assert lib.ds_op_describe(C.byref(op), nm, 128, C.byref(fl), C.byref(by)) == 0
        return nm.value.decode()

    assert name_of(M, N, K) == "gemm_g320_kernel<plain>"
    assert name_of(65536, N, K) != "gemm_g320_kernel<plain>" and name_of(2048, N, K) != "gemm_g320_kernel<plain>"
    got = ops.gemm(x, w, b)
    ref = F.linear(x.float().cpu(), w.float().cpu(), b.float().cpu())
    e = _relmax(got, ref)
    print(f"gemm_g320 plain M={M} N={N} K={K}: {e:.2e}")
    assert e <= 2e-3, e
    gamma, beta = (1 + 0.2 * torch.randn(K, generator=g)).half().to(DEV), _r((K,), g, 0.2).to(DEV)
    gw, c2, b2 = pack_ln_fused(w, None, gamma, beta)
    xs = x.float().view(M, K // 64, 64)
    part = torch.stack([xs.sum(-1).t(), (xs * xs).sum(-1).t()], dim=-1).contiguous()
    got_ln = ops.gemm_ln_partial(x, gw, b2, c2, part)
    ref_ln = F.linear(F.layer_norm(x.float().cpu(), (K,), gamma.float().cpu(), beta.float().cpu(), 1e-5), w.float().cpu())
    e2 = _relmax(got_ln, ref_ln)
    print(f"gemm_g320 plain, fused-LayerNorm consumer: {e2:.2e}")
    assert e2 <= 3e-3, e2
    assert lib.ds_set_option(b"gemm_g320", 1) == 0
    try:
        assert name_of(M, N, K) != "gemm_g320_kernel<plain>"
        assert torch.equal(ops.gemm(x, w, b), got) and torch.equal(ops.gemm_ln_partial(x, gw, b2, c2, part), got_ln)
    finally:
        lib.ds_set_option(b"gemm_g320", 0)
