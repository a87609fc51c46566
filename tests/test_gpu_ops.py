"""GPU: every HIP kernel through the C ABI vs a plain PyTorch fp32 reference of the same op (computed on the CPU
from the same fp16-rounded inputs) and, where the reference's own code produced a fixture, vs that fixture.

Tolerances (stated per test): fp16 outputs of fp32-accumulated contractions -> max |err| <= 2e-3 * max|ref|
(one fp16 ulp is 4.9e-4 relative); softmax-weighted sums -> 2e-3 absolute on O(1) data; index/mask work -> bit-exact.
"""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ops(hip_lib):
    from diffsensei_amd import ops
    return ops


def _r(shape, g, scale=1.0):
    return (torch.randn(shape, generator=g) * scale).half()


def _close(got, ref, tol=2e-3, what=""):
    got, ref = got.float().cpu(), ref.float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), what
    err = (got - ref).abs().max().item()
    den = max(ref.abs().max().item(), 1e-3)
    assert err <= tol * den + 1e-3 * tol, f"{what}: max err {err:.4g} vs max|ref| {den:.4g}"


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 320), (200, 136, 192), (2048, 1280, 1280),
                                   (8192, 640, 2560), (77, 96, 2048), (4, 1280, 768), (1000, 5120, 640)])
def test_gemm_bias_residual(hip_lib, M, N, K):
    ops = _ops(hip_lib)
    g = torch.Generator().manual_seed(M + N + K)
    x, w, b, r = _r((M, K), g), _r((N, K), g, 1 / math.sqrt(K)), _r((N,), g), _r((M, N), g)
    ref = x.float() @ w.float().t() + b.float()
    _close(ops.gemm(x.to(DEV), w.to(DEV)), x.float() @ w.float().t(), what="plain")
    _close(ops.gemm(x.to(DEV), w.to(DEV), b.to(DEV)), ref, what="bias")
    ref_r = ref.half().float() + r.float()
    _close(ops.gemm(x.to(DEV), w.to(DEV), b.to(DEV), r.to(DEV)), ref_r, what="bias+residual")


def test_gemm_asymmetric_layout(hip_lib):
    """Transpose-detecting check (asymmetric operands): A = row/col pattern, W = one-hot rows."""
    ops = _ops(hip_lib)
    M, N, K = 192, 256, 128
    x = (torch.arange(M)[:, None] * 0.01 + torch.arange(K)[None, :] * 0.1).half()
    w = torch.zeros(N, K).half()
    for n in range(N):
        w[n, (n * 7) % K] = 1.0 + (n % 3)
    ref = x.float() @ w.float().t()
    _close(ops.gemm(x.to(DEV), w.to(DEV)), ref, tol=1e-3, what="asymmetric")


def test_gemm_inplace_residual(hip_lib):
    ops = _ops(hip_lib)
    g = torch.Generator().manual_seed(5)
    M, N, K = 512, 640, 640
    x, w, b, r = _r((M, K), g), _r((N, K), g, 1 / math.sqrt(K)), _r((N,), g), _r((M, N), g)
    ref = (x.float() @ w.float().t() + b.float()).half().float() + r.float()
    rd = r.to(DEV)
    out = ops.gemm(x.to(DEV), w.to(DEV), b.to(DEV), residual=rd, out=rd)
    _close(out, ref, what="in-place residual")


def test_gemm_split_a(hip_lib):
    ops = _ops(hip_lib)
    g = torch.Generator().manual_seed(6)
    M, N, K1, K2 = 300, 320, 640, 320
    x1, x2, w = _r((M, K1), g), _r((M, K2), g), _r((N, K1 + K2), g, 1 / math.sqrt(K1 + K2))
    ref = torch.cat([x1, x2], 1).float() @ w.float().t()
    _close(ops.gemm(x1.to(DEV), w.to(DEV), x2=x2.to(DEV)), ref, what="split A")


@pytest.mark.parametrize("act", ["gelu", "quick_gelu"])
def test_gemm_activation(hip_lib, act):
    ops = _ops(hip_lib)
    g = torch.Generator().manual_seed(7)
    M, N, K = 260, 512, 256
    x, w, b = _r((M, K), g), _r((N, K), g, 1 / math.sqrt(K)), _r((N,), g)
    y = (x.float() @ w.float().t() + b.float()).half().float()
    ref = F.gelu(y) if act == "gelu" else y * torch.sigmoid(1.702 * y)
    _close(ops.gemm(x.to(DEV), w.to(DEV), b.to(DEV), act=act), ref, what=act)


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (512, 768, 384), (272, 640, 256), (1040, 384, 640),
                                   (4144, 1408, 640), (16, 128, 128), (8192, 2560, 1280)])
def test_gemm_pingpong_kernel(hip_lib, M, N, K):
    """256x256 persistent ping-pong kernel (gemm_pp.hip), forced with gemm_variant 3: fp32 reference, and bit-exact
    agreement with the register-staged kernel (same MFMA order per output).  Ragged M/N (multiples of 16) included."""
    from diffsensei_amd import _lib
    ops = _ops(hip_lib)
    lib = _lib.load()
    g = torch.Generator().manual_seed(M * 3 + N + K)
    x, w, b, r = _r((M, K), g), _r((N, K), g, 1 / math.sqrt(K)), _r((N,), g), _r((M, N), g)
    ref = (x.float() @ w.float().t() + b.float()).half().float() + r.float()
    try:
        assert lib.ds_set_option(b"gemm_variant", 3) == 0
        got = ops.gemm(x.to(DEV), w.to(DEV), b.to(DEV), r.to(DEV))
        got_act = ops.gemm(x.to(DEV), w.to(DEV), b.to(DEV), act="gelu")
        assert lib.ds_set_option(b"gemm_variant", 1) == 0
        base = ops.gemm(x.to(DEV), w.to(DEV), b.to(DEV), r.to(DEV))
    finally:
        lib.ds_set_option(b"gemm_variant", 0)
    _close(got, ref, what="pingpong bias+residual")
    assert torch.equal(got, base)
    _close(got_act, F.gelu(x.float() @ w.float().t() + b.float()), what="pingpong gelu")


def test_gemm_pingpong_geglu_and_dispatch(hip_lib):
    """GEGLU epilogue of the ping-pong kernel (hidden/gate pairing in registers) on a ragged packed width, and the
    automatic dispatch picking it for a shape that fills whole rounds of CUs."""
    from diffsensei_amd import _lib
    from diffsensei_amd.engine import pack_geglu
    ops = _ops(hip_lib)
    lib = _lib.load()
    g = torch.Generator().manual_seed(11)
    M, C = 1040, 176  # packed N = 8*C = 1408 = 5.5 tiles of 256
    x, w, b = _r((M, 384), g), _r((8 * C, 384), g, 1 / math.sqrt(384)), _r((8 * C,), g)
    p = (x.float() @ w.float().t() + b.float()).half().float()
    hid, gate = p.chunk(2, dim=-1)
    ref = hid * F.gelu(gate).half().float()
    wp, bp = pack_geglu(w, b)
    try:
        assert lib.ds_set_option(b"gemm_variant", 3) == 0
        got = ops.gemm(x.to(DEV), wp.to(DEV), bp.to(DEV), geglu=True)
    finally:
        lib.ds_set_option(b"gemm_variant", 0)
    _close(got, ref, what="pingpong geglu")
    # auto dispatch: 32 x 40 tiles = 5 full rounds -> ping-pong; results must not depend on the choice
    x2, w2 = _r((8192, 1280), g), _r((10240, 1280), g, 1 / math.sqrt(1280))
    auto = ops.gemm(x2.to(DEV), w2.to(DEV))
    try:
        lib.ds_set_option(b"gemm_variant", 8)
        forced = ops.gemm(x2.to(DEV), w2.to(DEV))
    finally:
        lib.ds_set_option(b"gemm_variant", 0)
    assert torch.equal(auto, forced)
    _close(auto, x2.float() @ w2.float().t(), what="auto dispatch")


@pytest.mark.parametrize("M,N,K,mode", [(32768 - 48, 1280, 1280, "res"), (16384, 2560 - 64, 1280, "bias"),
                                        (24576, 1280, 256, "inplace"), (16384, 2560, 1280, "none"),
                                        (8192, 10240, 1280, "geglu"), (20480, 1280, 640, "gelu")])
def test_gemm_pingpong_tile_handover(hip_lib, M, N, K, mode):
    """Several output tiles per persistent block (> 256 tiles), so every hand-over path of gemm_pp_kernel runs: the bias
    slice staged through LDS, residual rows requested a piece ahead (also in place: C == residual), interior tiles next
    to ragged ones (branch-free and generic epilogues alternate inside one block's walk, `pad_tail`), no bias at all,
    an activation (generic epilogue everywhere).  The C stores of a tile drain under the next tile's first k-tiles
    (counted waits): every launch must equal, bit for bit, the variant that drains them first (gemm_debug 256) - an
    under-counted wait shows up as a rare wrong tile, hence the repetitions - and agree with fp32."""
    from diffsensei_amd import _lib
    from diffsensei_amd.engine import pack_geglu
    ops = _ops(hip_lib)
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    x, w = _r((M, K), g).to(DEV), _r((N, K), g, 1 / math.sqrt(K)).to(DEV)
    b = None if mode == "none" else _r((N,), g).to(DEV)
    res = _r((M, N), g).to(DEV) if mode in ("res", "inplace") else None
    kw = {}
    if mode == "geglu":
        w, b = pack_geglu(w, b)
        kw["geglu"] = True
    if mode == "gelu":
        kw["act"] = "gelu"

    def run(out=None):
        if mode == "inplace":
            buf = res.clone()
            return ops.gemm(x, w, b, residual=buf, out=buf)
        return ops.gemm(x, w, b, residual=res, out=out, **kw)

    try:
        assert lib.ds_set_option(b"gemm_variant", 3) == 0
        assert lib.ds_set_option(b"gemm_debug", 256) == 0
        drained = run().clone()
        assert lib.ds_set_option(b"gemm_debug", 0) == 0
        for _ in range(10):
            assert torch.equal(run(), drained)
    finally:
        lib.ds_set_option(b"gemm_debug", 0)
        lib.ds_set_option(b"gemm_variant", 0)
    y = x.float() @ w.float().t()
    if b is not None:
        y = y + b.float()
    if mode == "geglu":  # packed layout: every 128 columns = 64 hidden | their 64 gates
        t = y.half().float().view(M, N // 128, 2, 64)
        ref = (t[:, :, 0] * F.gelu(t[:, :, 1]).half().float()).reshape(M, N // 2)
    elif mode == "gelu":
        ref = F.gelu(y)
    else:
        ref = y.half().float() + (res.float() if res is not None else 0)
    _close(drained, ref, what=f"pingpong hand-over {mode}")


@pytest.mark.parametrize("M,N,K,mode", [(32768, 640, 640, "res"), (24576, 640, 640, "bias"), (16384, 320, 256, "none"),
                                        (1024, 192, 128, "res"), (65536, 640, 2560, "res")])
def test_gemm_pingpong_ragged_column_strips(hip_lib, M, N, K, mode):
    """Round 6: the ragged last tile COLUMN of an N % 64 == 0 problem (the 640-channel level: 2.5 tile columns) runs the
    branch-free epilogue wave by wave - a wave's 64 columns are wholly inside N or the wave stores nothing and only pads its
    vector-memory count (gemm_pp.hip `strips_ok` / `wave_cols_in`).  Bit-identical to the same tiles on the generic epilogue
    (gemm_debug 4096), to the drained hand-over (256) and to the 128 x 128 kernel; repeated, because a mis-counted wait shows up
    as a rare wrong tile; the columns behind N in the (wider) output buffer must stay untouched; the automatic dispatch now
    gives the N = K = 640 projection at a large M to this kernel (knob gemm_pp_narrow 1: the 128 x 128 kernels, same bits)."""
    from diffsensei_amd import _lib
    ops = _ops(hip_lib)
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + N + K + 1)
    x, w = _r((M, K), g).to(DEV), _r((N, K), g, 1 / math.sqrt(K)).to(DEV)
    b = None if mode == "none" else _r((N,), g).to(DEV)
    res = _r((M, N), g).to(DEV) if mode == "res" else None
    try:
        assert lib.ds_set_option(b"gemm_variant", 3) == 0
        assert lib.ds_set_option(b"gemm_debug", 4096) == 0
        generic = ops.gemm(x, w, b, residual=res).clone()
        assert lib.ds_set_option(b"gemm_debug", 256) == 0
        drained = ops.gemm(x, w, b, residual=res).clone()
        assert lib.ds_set_option(b"gemm_debug", 0) == 0
        for _ in range(10):
            assert torch.equal(ops.gemm(x, w, b, residual=res), generic)
        assert torch.equal(drained, generic)
        # a wider output buffer: the waves outside N must not store (ldc = N + 128, canary behind the columns)
        wide = torch.full((M, N + 128), 7.0, dtype=torch.float16, device=DEV)
        rc = lib.ds_gemm_f16(ops._p(x), K, None, 0, K, ops._p(w), K, ops._p(b), ops._p(res), N, ops._p(wide), N + 128, M, N, K, 0,
                             ops._stream())
        _lib.check(rc, "ds_gemm_f16 (ldy = N + 128)")
        assert torch.equal(wide[:, :N], generic) and bool((wide[:, N:] == 7.0).all())
        assert lib.ds_set_option(b"gemm_variant", 8) == 0
        assert torch.equal(ops.gemm(x, w, b, residual=res), generic)
    finally:
        lib.ds_set_option(b"gemm_debug", 0)
        lib.ds_set_option(b"gemm_variant", 0)
    ref = x.float() @ w.float().t()
    if b is not None:
        ref = ref + b.float()
    ref = ref.half().float() + (res.float() if res is not None else 0)
    _close(generic, ref, what="pingpong ragged column strips")
    if (M, N, K) == (32768, 640, 640):
        from diffsensei_amd.ops import gemm_ln_fusable
        assert gemm_ln_fusable(65536, 640, 640) == 1 and gemm_ln_fusable(262144, 640, 640) == 1
        assert torch.equal(ops.gemm(x, w, b, residual=res), generic)     # automatic dispatch
        try:
            assert lib.ds_set_option(b"gemm_pp_narrow", 1) == 0
            assert gemm_ln_fusable(65536, 640, 640) == 2
            assert torch.equal(ops.gemm(x, w, b, residual=res), generic)
        finally:
            lib.ds_set_option(b"gemm_pp_narrow", 0)


@pytest.mark.parametrize("M,N,K", [(2048, 1280, 1280), (2048, 2560, 1280), (2048, 1280, 5120), (8192, 640, 640),
                                   (200, 136, 256), (64, 128, 320), (1000, 640, 2560), (4096, 1280, 1280)])
def test_gemm_ring_buffered_small_grid_kernel(hip_lib, M, N, K):
    """`gemm_glds_kernel<64,false,3|4>` (ring of LDS buffers, the automatic choice for grids of <= 512 blocks: the
    num_samples-1 shapes M = 2048 / 8192): vs fp32, and bit-identical to the one-buffer kernel it replaces
    (gemm_ring 1), with bias + residual, split A, GEGLU and the batched V^T form."""
    from diffsensei_amd import _lib
    from diffsensei_amd.engine import pack_geglu
    ops = _ops(hip_lib)
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + 3 * N + K)
    x, w, b, r = _r((M, K), g), _r((N, K), g, 1 / math.sqrt(K)), _r((N,), g), _r((M, N), g)
    ref = (x.float() @ w.float().t() + b.float()).half().float() + r.float()
    K1 = (K // 128) * 64
    out = {}
    try:
        for v in (0, 1):
            assert lib.ds_set_option(b"gemm_ring", v) == 0
            o = [ops.gemm(x.to(DEV), w.to(DEV), b.to(DEV), r.to(DEV)),
                 ops.gemm(x[:, :K1].contiguous().to(DEV), w.to(DEV), x2=x[:, K1:].contiguous().to(DEV))]
            if N % 128 == 0:
                wp, bp = pack_geglu(w, b)
                o.append(ops.gemm(x.to(DEV), wp.to(DEV), bp.to(DEV), geglu=True))
            out[v] = o
    finally:
        lib.ds_set_option(b"gemm_ring", 0)
    _close(out[0][0], ref, what="ring bias+residual")
    _close(out[0][1], x.float() @ w.float().t(), what="ring split A")
    for a, c in zip(out[0], out[1]):
        assert torch.equal(a, c), "ring-buffered and one-buffer kernels differ"


@pytest.mark.parametrize("M,C", [(256, 128), (2048, 640), (777, 256)])
def test_gemm_geglu(hip_lib, M, C):
    from diffsensei_amd.engine import pack_geglu
    ops = _ops(hip_lib)
    g = torch.Generator().manual_seed(M + C)
    x, w, b = _r((M, C), g), _r((8 * C, C), g, 1 / math.sqrt(C)), _r((8 * C,), g)
    p = (x.float() @ w.float().t() + b.float()).half().float()
    hid, gate = p.chunk(2, dim=-1)
    ref = hid * F.gelu(gate).half().float()
    wp, bp = pack_geglu(w, b)
    _close(ops.gemm(x.to(DEV), wp.to(DEV), bp.to(DEV), geglu=True), ref, what="geglu")


def test_gemm_batched_vt(hip_lib):
    ops = _ops(hip_lib)
    g = torch.Generator().manual_seed(8)
    B, N, C = 3, 200, 256
    x, wv = _r((B, N, C), g), _r((C, C), g, 1 / math.sqrt(C))
    ref = torch.einsum("ck,bnk->bcn", wv.float(), x.float())
    _close(ops.gemm_batched_nt(wv.to(DEV), x.to(DEV)), ref, what="V^T")


@pytest.mark.parametrize("B,N,C", [(32, 1024, 1280), (8, 4096, 640), (16, 1024, 1280), (5, 1000, 1280), (3, 272, 384)])
def test_gemm_batched_vt_pingpong_folded_batch(hip_lib, B, N, C):
    """Round 3: batched problems (V^T[b] = Wv X_b^T, A shared) are folded into gemm_pp_kernel's persistent tile walk
    (id -> image, tile).  The UNet's level-2 / level-1 shapes by natural dispatch and forced (gemm_variant 3), incl. a batch
    whose tiles are not a multiple of the grid and ragged M / N: vs fp32, and bit-identical to the register-staged kernel."""
    from diffsensei_amd import _lib
    ops = _ops(hip_lib)
    lib = _lib.load()
    g = torch.Generator().manual_seed(B + N + C)
    x, wv = _r((B, N, C), g).to(DEV), _r((C, C), g, 1 / math.sqrt(C)).to(DEV)
    ref = torch.einsum("ck,bnk->bcn", wv.float().cpu(), x.float().cpu())
    out = {}
    try:
        for var in (1, 3, 0):
            assert lib.ds_set_option(b"gemm_variant", var) == 0
            out[var] = ops.gemm_batched_nt(wv, x).clone()
    finally:
        lib.ds_set_option(b"gemm_variant", 0)
    _close(out[3], ref, what="batched V^T ping-pong")
    assert torch.equal(out[3], out[1]) and torch.equal(out[0], out[1])
    # poisoned output buffer: every element of every image is written exactly by its own tile
    buf = torch.full((B, C, N), float("nan"), dtype=torch.float16, device=DEV)
    lib.ds_set_option(b"gemm_variant", 3)
    try:
        ops.gemm_batched_nt(wv, x, out=buf)
    finally:
        lib.ds_set_option(b"gemm_variant", 0)
    assert torch.equal(buf, out[1])


# ------------------------------------------------------------------------------------------------ conv
@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,up", [(2, 16, 16, 64, 128, 1, False), (1, 32, 24, 128, 64, 1, False),
                                                       (2, 16, 16, 64, 64, 2, False), (2, 8, 12, 128, 128, 1, True),
                                                       (2, 64, 64, 320, 320, 1, False), (1, 32, 32, 1920, 640, 1, False),
                                                       (2, 8, 16, 128, 192, 1, True), (3, 24, 48, 64, 320, 1, False),
                                                       (2, 20, 24, 64, 128, 1, False), (1, 9, 13, 64, 64, 1, True)])
def test_conv3x3(hip_lib, B, H, W, Cin, Cout, stride, up):
    ops = _ops(hip_lib)
    g = torch.Generator().manual_seed(B * H + Cin + Cout + stride)
    x = _r((B, Cin, H, W), g)
    w, b = _r((Cout, Cin, 3, 3), g, 1 / math.sqrt(9 * Cin)), _r((Cout,), g)
    xi = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if up else x.float()
    ref = F.conv2d(xi, w.float(), b.float(), stride=stride, padding=1)
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    w_p = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    y = ops.conv3x3(x_nhwc, w_p, b.to(DEV), stride=stride, upsample=up)
    _close(y.permute(0, 3, 1, 2), ref, what="conv")
    # + per-image bias (time embedding) + residual
    rb, res = _r((B, Cout), g), _r(tuple(ref.shape), g)
    ref2 = (ref + rb.float()[:, :, None, None]).half().float() + res.float()
    y2 = ops.conv3x3(x_nhwc, w_p, b.to(DEV), stride=stride, upsample=up, rowbias=rb.to(DEV),
                     residual=res.permute(0, 2, 3, 1).contiguous().to(DEV))
    _close(y2.permute(0, 3, 1, 2), ref2, what="conv+rowbias+res")


def _conv_case(g, B, H, W, Cin, Cout, up, dtype=torch.float16):
    x = (torch.randn((B, Cin, H, W), generator=g)).to(dtype)
    w = (torch.randn((Cout, Cin, 3, 3), generator=g) / math.sqrt(9 * Cin)).to(dtype)
    b = torch.randn((Cout,), generator=g).to(dtype)
    xi = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if up else x.float()
    ref = F.conv2d(xi, w.float(), b.float(), padding=1)
    return x.permute(0, 2, 3, 1).contiguous().to(DEV), w.permute(0, 2, 3, 1).contiguous().to(DEV), b.to(DEV), ref


@pytest.mark.parametrize("B,H,W,Cin,Cout,up", [(2, 32, 32, 64, 128, False), (1, 20, 24, 128, 192, False),
                                                (2, 9, 13, 64, 64, True), (1, 64, 64, 320, 320, False),
                                                (3, 16, 48, 640, 320, False), (1, 24, 40, 1280, 640, True)])
def test_conv_halo256_forced_variant(hip_lib, B, H, W, Cin, Cout, up):
    """`conv_halo256_kernel<half>` (16x16-pixel blocks; the variant the benchmark's batch-32 forward dispatches to) forced
    with conv_halo_variant 2 on shapes whose grids are far below its automatic threshold, incl. ragged edge patches and the
    fused x2 upsample: vs F.conv2d in fp32, and bit-identical to `conv_halo_kernel` (8x16 pixels, variant 1) - and so is
    `conv_halo_deep_kernel` (variant 3: the ring-buffered 8x16 kernel small grids dispatch to since round 6; one to twenty
    channel slices, ragged patches, the fused upsample, per-image bias + residual)."""
    from diffsensei_amd import _lib
    ops = _ops(hip_lib)
    lib = _lib.load()
    g = torch.Generator().manual_seed(B * H + Cin + Cout + int(up))
    x, w, b, ref = _conv_case(g, B, H, W, Cin, Cout, up)
    rb = _r((B, Cout), g)
    res = _r(tuple(ref.shape), g)
    ref2 = (ref + rb.float()[:, :, None, None]).half().float() + res.float()
    res_d = res.permute(0, 2, 3, 1).contiguous().to(DEV)
    out = {}
    try:
        for v in (1, 2, 3, 0):
            assert lib.ds_set_option(b"conv_halo_variant", v) == 0
            out[v] = (ops.conv3x3(x, w, b, upsample=up),
                      ops.conv3x3(x, w, b, upsample=up, rowbias=rb.to(DEV), residual=res_d))
    finally:
        lib.ds_set_option(b"conv_halo_variant", 0)
    _close(out[2][0].permute(0, 3, 1, 2), ref, what="halo256")
    _close(out[2][1].permute(0, 3, 1, 2), ref2, what="halo256+rowbias+res")
    _close(out[3][0].permute(0, 3, 1, 2), ref, what="halo deep")
    assert torch.equal(out[1][0], out[2][0]) and torch.equal(out[1][1], out[2][1]), "16x16 and 8x16 halo kernels differ"
    assert torch.equal(out[1][0], out[3][0]) and torch.equal(out[1][1], out[3][1]), "ring-buffered and single-buffer 8x16 kernels differ"
    assert torch.equal(out[1][0], out[0][0]) and torch.equal(out[1][1], out[0][1]), "automatic dispatch differs"


@pytest.mark.parametrize("B,H,W,Cin,Cout,up", [(8, 128, 128, 320, 320, False), (16, 32, 32, 640, 640, True)])
def test_conv_halo256_natural_dispatch(hip_lib, B, H, W, Cin, Cout, up):
    """Shapes of the benchmarked forward (batch 32: B x 128 x 128 x 320 -> 320; here B = 8 -> 1536 blocks >= 1024, and
    the level-1 upsampler) that reach `conv_halo256_kernel` through the automatic dispatch of ds_conv3x3_f16."""
    from diffsensei_amd import _lib
    ops = _ops(hip_lib)
    lib = _lib.load()
    Ho, Wo = (2 * H, 2 * W) if up else (H, W)
    assert B * ((Ho + 15) // 16) * ((Wo + 15) // 16) * ((Cout + 127) // 128) >= 1024
    g = torch.Generator().manual_seed(B + H + Cin)
    x, w, b, ref = _conv_case(g, B, H, W, Cin, Cout, up)
    auto = ops.conv3x3(x, w, b, upsample=up)
    try:
        lib.ds_set_option(b"conv_halo_variant", 1)
        small = ops.conv3x3(x, w, b, upsample=up)
    finally:
        lib.ds_set_option(b"conv_halo_variant", 0)
    _close(auto.permute(0, 3, 1, 2), ref, what="halo256 auto")
    assert torch.equal(auto, small)


@pytest.mark.parametrize("B,H,W,Cin,Cout,up", [(1, 32, 32, 128, 128, False), (1, 24, 40, 256, 128, True),
                                                (2, 64, 64, 512, 256, False)])
def test_conv_halo256_bf16(hip_lib, B, H, W, Cin, Cout, up):
    """bf16 twin (VAE decoder) of the 16x16-pixel halo kernel: forced, vs fp32 conv and bit-identical to the 8x16 kernel."""
    from diffsensei_amd import _lib
    ops = _ops(hip_lib)
    lib = _lib.load()
    g = torch.Generator().manual_seed(B * H + Cin + Cout + 5)
    x, w, b, ref = _conv_case(g, B, H, W, Cin, Cout, up, dtype=torch.bfloat16)
    out = {}
    try:
        for v in (1, 2):
            assert lib.ds_set_option(b"conv_halo_variant", v) == 0
            out[v] = ops.conv3x3_bf16(x, w, b, upsample=up)
    finally:
        lib.ds_set_option(b"conv_halo_variant", 0)
    _close(out[2].permute(0, 3, 1, 2), ref, tol=1e-2, what="halo256 bf16")   # bf16 output: 8 mantissa bits
    assert torch.equal(out[1], out[2])


@pytest.mark.parametrize("B,H,W,Ho,Wo,Cin,Cout", [(2, 5, 4, 9, 7, 128, 64), (1, 9, 7, 18, 13, 64, 128), (2, 16, 16, 31, 32, 64, 64),
                                                   (1, 17, 33, 33, 65, 128, 192), (3, 8, 8, 16, 16, 64, 64)])
def test_conv3x3_resize_to_explicit_size(hip_lib, B, H, W, Ho, Wo, Cin, Cout):
    """Upsample2D with `output_size` (diffusers `forward_upsample_size`: latent sides that are not multiples of 4): nearest
    resize to an explicit size with F.interpolate's index rule, then the 3x3 conv - fused, in every conv kernel family
    (halo 8x16, halo 16x16, LDS-DMA gather, register-staged) with bit-identical results between the two halo kernels."""
    from diffsensei_amd import _lib
    ops = _ops(hip_lib)
    lib = _lib.load()
    g = torch.Generator().manual_seed(H * W + Ho + Cin)
    x = _r((B, Cin, H, W), g)
    w, b = _r((Cout, Cin, 3, 3), g, 1 / math.sqrt(9 * Cin)), _r((Cout,), g)
    ref = F.conv2d(F.interpolate(x.float(), size=(Ho, Wo), mode="nearest"), w.float(), b.float(), padding=1)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wd = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    outs = {}
    try:
        for name, opt, val in (("halo8x16", b"conv_halo_variant", 1), ("halo16x16", b"conv_halo_variant", 2),
                               ("glds", b"gemm_variant", 2), ("regs", b"gemm_variant", 1)):
            lib.ds_set_option(b"conv_halo_variant", 0)
            lib.ds_set_option(b"gemm_variant", 0)
            assert lib.ds_set_option(opt, val) == 0
            outs[name] = ops.conv3x3(xd, wd, b.to(DEV), out_size=(Ho, Wo))
    finally:
        lib.ds_set_option(b"conv_halo_variant", 0)
        lib.ds_set_option(b"gemm_variant", 0)
    for name, y in outs.items():
        assert y.shape == (B, Ho, Wo, Cout)
        _close(y.permute(0, 3, 1, 2), ref, what=f"resize conv {name}")
    assert torch.equal(outs["halo8x16"], outs["halo16x16"])
    if (Ho, Wo) == (2 * H, 2 * W):      # the explicit size 2H x 2W is the plain upsample
        assert torch.equal(outs["halo8x16"], ops.conv3x3(xd, wd, b.to(DEV), upsample=True))


def test_conv_in_dialog_and_conv_out(hip_lib):
    ops = _ops(hip_lib)
    g = torch.Generator().manual_seed(11)
    B, H, W, C = 2, 16, 24, 64
    x, w, b, emb = _r((B, 4, H, W), g), _r((C, 4, 3, 3), g, 1 / 6), _r((C,), g), _r((C,), g)
    boxes = torch.tensor([[[1, 2, 9, 7], [5, 5, 20, 16], [0, 0, 0, 0]], [[0, 0, 0, 0]] * 3], dtype=torch.int32)
    ref = F.conv2d(x.float(), w.float(), b.float(), padding=1).half().float()
    mask = torch.zeros(B, 1, H, W)
    for i in range(B):
        for (x1, y1, x2, y2) in boxes[i].tolist():
            mask[i, :, y1:y2, x1:x2] = 1
    ref = ref + mask * emb.float()[None, :, None, None]
    y = ops.conv_in_dialog(x.permute(0, 2, 3, 1).contiguous().to(DEV), w.permute(0, 2, 3, 1).contiguous().to(DEV),
                           b.to(DEV), boxes.to(DEV), emb.to(DEV))
    _close(y.permute(0, 3, 1, 2), ref, tol=1e-3, what="conv_in+dialog")
    xo, wo, bo = _r((B, C, H, W), g), _r((4, C, 3, 3), g, 1 / math.sqrt(9 * C)), _r((4,), g)
    ref = F.conv2d(xo.float(), wo.float(), bo.float(), padding=1)
    y = ops.conv_out(xo.permute(0, 2, 3, 1).contiguous().to(DEV), wo.permute(0, 2, 3, 1).contiguous().to(DEV), bo.to(DEV))
    _close(y.permute(0, 3, 1, 2), ref, what="conv_out")


def test_conv_in_out_grid_stride_sizes(hip_lib):
    """Sizes beyond the 2048-block cap of conv_in / conv_out (batch-32 shapes walk pixels grid-stride so the weight panel is
    staged once per resident block): same numerics, dialog boxes included."""
    ops = _ops(hip_lib)
    g = torch.Generator().manual_seed(12)
    B, H, W, C = 6, 128, 96, 64
    x, w, b, emb = _r((B, 4, H, W), g), _r((C, 4, 3, 3), g, 1 / 6), _r((C,), g), _r((C,), g)
    boxes = torch.zeros(B, 2, 4, dtype=torch.int32)
    boxes[1, 0] = torch.tensor([3, 5, 60, 40])
    boxes[5, 1] = torch.tensor([70, 100, 96, 128])
    ref = F.conv2d(x.float(), w.float(), b.float(), padding=1).half().float()
    mask = torch.zeros(B, 1, H, W)
    for i in range(B):
        for (x1, y1, x2, y2) in boxes[i].tolist():
            mask[i, :, y1:y2, x1:x2] = 1
    ref = ref + mask * emb.float()[None, :, None, None]
    y = ops.conv_in_dialog(x.permute(0, 2, 3, 1).contiguous().to(DEV), w.permute(0, 2, 3, 1).contiguous().to(DEV),
                           b.to(DEV), boxes.to(DEV), emb.to(DEV))
    assert B * H * W * (C // 8) // 256 > 2048
    _close(y.permute(0, 3, 1, 2), ref, tol=1e-3, what="conv_in grid-stride")
    xo, wo, bo = _r((B, C, H, W), g), _r((4, C, 3, 3), g, 1 / math.sqrt(9 * C)), _r((4,), g)
    ref = F.conv2d(xo.float(), wo.float(), bo.float(), padding=1)
    y = ops.conv_out(xo.permute(0, 2, 3, 1).contiguous().to(DEV), wo.permute(0, 2, 3, 1).contiguous().to(DEV), bo.to(DEV))
    assert B * H * W // 32 > 2048
    _close(y.permute(0, 3, 1, 2), ref, what="conv_out grid-stride")


# ------------------------------------------------------------------------------------------------ norms
@pytest.mark.parametrize("B,HW,C1,C2,silu,eps", [(2, 256, 64, 0, True, 1e-5), (2, 1024, 320, 0, True, 1e-5),
                                                  (1, 333, 128, 64, True, 1e-5), (2, 4096, 640, 0, False, 1e-6),
                                                  (2, 1024, 1280, 1280, True, 1e-5), (1, 64, 1920, 0, True, 1e-5)])
def test_groupnorm(hip_lib, B, HW, C1, C2, silu, eps):
    ops = _ops(hip_lib)
    g = torch.Generator().manual_seed(HW + C1 + C2)
    x1 = _r((B, HW, C1), g, 2.0) + 0.5
    x2 = _r((B, HW, C2), g) if C2 else None
    C = C1 + C2
    gamma, beta = (1 + 0.1 * torch.randn(C, generator=g)).half(), (0.1 * torch.randn(C, generator=g)).half()
    xc = torch.cat([x1, x2], -1) if C2 else x1
    ref = F.group_norm(xc.float().permute(0, 2, 1), 32, gamma.float(), beta.float(), eps).permute(0, 2, 1)
    if silu:
        ref = F.silu(ref)
    y = ops.groupnorm(x1.to(DEV), gamma.to(DEV), beta.to(DEV), 32, eps, silu, None if x2 is None else x2.to(DEV))
    _close(y, ref, tol=3e-3, what="groupnorm")


@pytest.mark.parametrize("rows,C", [(1000, 640), (2048, 1280), (37, 128), (64, 2048), (5, 768), (64, 5120), (9, 8192)])
def test_layernorm(hip_lib, rows, C):
    ops = _ops(hip_lib)
    g = torch.Generator().manual_seed(rows + C)
    x = _r((rows, C), g, 3.0) + 1.0
    gamma, beta = (1 + 0.1 * torch.randn(C, generator=g)).half(), (0.1 * torch.randn(C, generator=g)).half()
    ref = F.layer_norm(x.float(), (C,), gamma.float(), beta.float(), 1e-5)
    _close(ops.layernorm(x.to(DEV), gamma.to(DEV), beta.to(DEV)), ref, what="layernorm")


# ------------------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("B,heads,N", [(1, 2, 256), (2, 10, 1024), (1, 4, 960), (2, 1, 64), (1, 20, 1024), (1, 2, 72)])
def test_self_attention(hip_lib, B, heads, N):
    ops = _ops(hip_lib)
    g = torch.Generator().manual_seed(B + heads + N)
    C = heads * 64
    q, k, v = _r((B, N, C), g), _r((B, N, C), g), _r((B, N, C), g)
    hs = lambda t: t.float().view(B, N, heads, 64).transpose(1, 2)
    ref = F.scaled_dot_product_attention(hs(q), hs(k), hs(v)).transpose(1, 2).reshape(B, N, C)
    vt = v.view(B, N, heads, 64).permute(0, 2, 3, 1).contiguous()
    y = ops.self_attention(q.to(DEV), k.to(DEV), vt.to(DEV), heads)
    _close(y, ref, tol=3e-3, what="self-attn")


def _sdpa_case(g, B, heads, N):
    C = heads * 64
    q, k, v = _r((B, N, C), g), _r((B, N, C), g), _r((B, N, C), g)
    hs = lambda t: t.float().view(B, N, heads, 64).transpose(1, 2)
    ref = F.scaled_dot_product_attention(hs(q), hs(k), hs(v)).transpose(1, 2).reshape(B, N, C)
    vt = v.view(B, N, heads, 64).permute(0, 2, 3, 1).contiguous()
    return q.to(DEV), k.to(DEV), vt.to(DEV), ref


@pytest.mark.parametrize("B,heads,N", [(1, 2, 256), (2, 10, 1024), (1, 4, 960), (1, 2, 72), (1, 3, 4096), (2, 5, 2312)])
def test_self_attention_64row_variant_forced(hip_lib, B, heads, N):
    """`self_attn_kernel<2>` (64 query rows per wave: the variant the benchmark runs at N = 4096) forced with
    attn_variant 2, incl. ragged query/key counts: vs fp32 SDPA, and bit-identical to `self_attn_kernel<1>`."""
    from diffsensei_amd import _lib
    ops = _ops(hip_lib)
    lib = _lib.load()
    g = torch.Generator().manual_seed(B + heads + N)
    q, k, vt, ref = _sdpa_case(g, B, heads, N)
    out = {}
    try:
        for v in (1, 2):
            assert lib.ds_set_option(b"attn_variant", v) == 0
            out[v] = ops.self_attention(q, k, vt, heads)
    finally:
        lib.ds_set_option(b"attn_variant", 0)
    _close(out[2], ref, tol=3e-3, what="self-attn<2>")
    assert torch.equal(out[1], out[2]), "64-row and 32-row flash kernels differ"


def test_self_attention_natural_dispatch(hip_lib):
    """What ds_self_attn_f16 picks by itself (round 5 rule: the software-pipelined kernel from 128 blocks of 256 query rows on,
    i.e. for every UNet shape at every batch; the 32-row flash kernel below): a level-1 shape class at UNet batch 4 (640
    blocks) must run `self_attn_sp_kernel` - same bits as attn_variant 3 -, a 16-block problem `self_attn_kernel<1>`."""
    from diffsensei_amd import _lib
    ops = _ops(hip_lib)
    lib = _lib.load()
    for (B, heads, N, forced) in ((4, 10, 4096, 3), (1, 4, 1024, 1)):
        blocks = ((N + 255) // 256) * B * heads
        assert (blocks >= 128) == (forced == 3)
        g = torch.Generator().manual_seed(77 + N)
        q, k, vt, ref = _sdpa_case(g, B, heads, N)
        auto = ops.self_attention(q, k, vt, heads)
        try:
            lib.ds_set_option(b"attn_variant", forced)
            same = ops.self_attention(q, k, vt, heads)
        finally:
            lib.ds_set_option(b"attn_variant", 0)
        _close(auto, ref, tol=3e-3, what=f"self-attn auto N={N}")
        assert torch.equal(auto, same), (B, heads, N)


@pytest.mark.parametrize("B,heads,N", [(2, 2, 63), (1, 4, 99), (1, 2, 1001), (2, 1, 20), (1, 3, 2317)])
def test_self_attention_any_token_count(hip_lib, B, heads, N):
    """Token counts that are not multiples of 8 (odd latent sizes): V^T rows padded to 8 keys with arbitrary finite
    content in the padding (here: large values, which must not leak), both kernel variants."""
    from diffsensei_amd import _lib
    ops = _ops(hip_lib)
    lib = _lib.load()
    g = torch.Generator().manual_seed(B + heads + N)
    C = heads * 64
    q, k, v = _r((B, N, C), g), _r((B, N, C), g), _r((B, N, C), g)
    hs = lambda t: t.float().view(B, N, heads, 64).transpose(1, 2)
    ref = F.scaled_dot_product_attention(hs(q), hs(k), hs(v)).transpose(1, 2).reshape(B, N, C)
    Np = (N + 7) // 8 * 8
    vt = torch.full((B, heads, 64, Np), 100.0, dtype=torch.float16)
    vt[..., :N] = v.view(B, N, heads, 64).permute(0, 2, 3, 1)
    out = {}
    try:
        for var in (1, 2):
            lib.ds_set_option(b"attn_variant", var)
            out[var] = ops.self_attention(q.to(DEV), k.to(DEV), vt.to(DEV), heads)
    finally:
        lib.ds_set_option(b"attn_variant", 0)
    _close(out[1], ref, tol=3e-3, what="ragged self-attn")
    assert torch.equal(out[1], out[2])


def test_self_attention_forced_rescale(hip_lib):
    """One key dominates late in the sequence: the running max jumps and every accumulator must be rescaled."""
    ops = _ops(hip_lib)
    g = torch.Generator().manual_seed(3)
    B, heads, N = 1, 1, 320
    q, k, v = _r((B, N, 64), g), _r((B, N, 64), g), _r((B, N, 64), g)
    k[0, 300] = q[0, 17] * 6.0
    ref = F.scaled_dot_product_attention(q.float()[:, None], k.float()[:, None], v.float()[:, None])[:, 0]
    vt = v.view(B, N, 1, 64).permute(0, 2, 3, 1).contiguous()
    _close(ops.self_attention(q.to(DEV), k.to(DEV), vt.to(DEV), 1), ref, tol=3e-3, what="rescale")


def _with_attn_variant(lib, var, fn):
    assert lib.ds_set_option(b"attn_variant", var) == 0
    try:
        return fn()
    finally:
        lib.ds_set_option(b"attn_variant", 0)


@pytest.mark.parametrize("B,heads,N", [(1, 2, 256), (2, 10, 1024), (1, 4, 960), (1, 2, 72), (1, 3, 4096), (2, 5, 2312),
                                       (1, 1, 64), (1, 2, 128), (1, 1, 40), (2, 2, 1000)])
def test_self_attention_software_pipelined_variant(hip_lib, B, heads, N):
    """`self_attn_sp_kernel` (attention_sp.hip, attn_variant 3: scores of pair k+1 and P V of pair k-1 on the matrix pipe while
    the VALU exponentiates pair k; deferred rescale; LDS-DMA ring) vs fp32 SDPA at the tolerance of the other flash kernels,
    incl. one-tile problems, ragged query / key counts, and agreement with `self_attn_kernel<1>` to fp16 rounding."""
    from diffsensei_amd import _lib
    ops = _ops(hip_lib)
    lib = _lib.load()
    g = torch.Generator().manual_seed(B + heads + N)
    q, k, vt, ref = _sdpa_case(g, B, heads, N)
    got = _with_attn_variant(lib, 3, lambda: ops.self_attention(q, k, vt, heads))
    base = _with_attn_variant(lib, 1, lambda: ops.self_attention(q, k, vt, heads))
    _close(got, ref, tol=3e-3, what="self-attn sp")
    _close(got, base.float().cpu(), tol=2e-3, what="self-attn sp vs <1>")
    # determinism: the ring / skew bookkeeping has no launch-to-launch freedom
    again = _with_attn_variant(lib, 3, lambda: ops.self_attention(q, k, vt, heads))
    assert torch.equal(got, again)


@pytest.mark.parametrize("B,heads,N", [(2, 2, 63), (1, 4, 99), (1, 2, 1001), (2, 1, 20), (1, 3, 2317)])
def test_self_attention_sp_any_token_count(hip_lib, B, heads, N):
    """Token counts that are not multiples of 8: V^T rows padded to 8 keys with LARGE finite content that must not leak
    (the kernel clamps its LDS-DMA chunks to the last one holding a real key and masks scores of keys >= Nk)."""
    from diffsensei_amd import _lib
    ops = _ops(hip_lib)
    lib = _lib.load()
    g = torch.Generator().manual_seed(B + heads + N)
    C = heads * 64
    q, k, v = _r((B, N, C), g), _r((B, N, C), g), _r((B, N, C), g)
    hs = lambda t: t.float().view(B, N, heads, 64).transpose(1, 2)
    ref = F.scaled_dot_product_attention(hs(q), hs(k), hs(v)).transpose(1, 2).reshape(B, N, C)
    Np = (N + 7) // 8 * 8
    vt = torch.full((B, heads, 64, Np), 100.0, dtype=torch.float16)
    vt[..., :N] = v.view(B, N, heads, 64).permute(0, 2, 3, 1)
    got = _with_attn_variant(lib, 3, lambda: ops.self_attention(q.to(DEV), k.to(DEV), vt.to(DEV), heads))
    _close(got, ref, tol=3e-3, what="ragged self-attn sp")


@pytest.mark.parametrize("spike_at,gain", [(300, 6.0), (70, 10.0), (5, 8.0), (1000, 4.0)])
def test_self_attention_sp_deferred_rescale(hip_lib, spike_at, gain):
    """The deferred rescale (since round 5: a row is re-centred when a lane's 32-key partial sum of f16 probabilities exceeds
    2^14 - or overflowed to inf) is a rare data-dependent branch: force it.  One key aligned with
    one query row makes that row's maximum jump by far more than the threshold at a chosen tile (first, second, late);
    a second case scales ALL scores so that many rows cross the threshold at different tiles, and a third keeps every score
    far BELOW the initial reference (the first tile must re-centre, or the f16 probabilities would underflow)."""
    from diffsensei_amd import _lib
    ops = _ops(hip_lib)
    lib = _lib.load()
    g = torch.Generator().manual_seed(3 + spike_at)
    B, heads, N = 1, 2, 1088
    C = heads * 64
    q, k, v = _r((B, N, C), g), _r((B, N, C), g), _r((B, N, C), g)
    k[0, spike_at, :64] = q[0, 17, :64] * gain
    k[0, min(spike_at + 400, N - 1), 64:] = q[0, 700, 64:] * gain
    hs = lambda t: t.float().view(B, N, heads, 64).transpose(1, 2)
    vt = v.view(B, N, heads, 64).permute(0, 2, 3, 1).contiguous()
    for name, qq in (("spike", q), ("hot", q * 5.0), ("cold", q)):
        kk = k if name != "cold" else -(q.abs() + 1.0) * torch.sign(q) * 3.0   # q . k = -3 sum(q^2 + |q|): every score << 0
        ref = F.scaled_dot_product_attention(hs(qq), hs(kk), hs(v)).transpose(1, 2).reshape(B, N, C)
        got = _with_attn_variant(lib, 3, lambda: ops.self_attention(qq.to(DEV), kk.to(DEV), vt.to(DEV), heads))
        assert torch.isfinite(got).all(), name
        # "hot": logits of +-40; Q is pre-multiplied by scale * log2(e) and rounded to f16 once, so the logit error grows with
        # |logit| (2^-11 relative) - 8e-3 there, the flash kernels' usual 4e-3 otherwise
        _close(got, ref, tol=8e-3 if name == "hot" else 4e-3, what=f"deferred rescale [{name}]")


def test_self_attention_sp_matches_reference_fixture(hip_lib, golden_dir):
    """tests/golden/self_attn.npz is the output of the reference's own AttnProcessor2_0 (oracle/make_golden.py imports
    src/models/attention_processor.py unmodified): the product's processor with the software-pipelined kernel forced."""
    from diffsensei_amd import _lib
    from diffsensei_amd.attention_processor import AttentionWeights, AttnProcessor2_0
    lib = _lib.load()
    g2 = np.load(os.path.join(golden_dir, "self_attn.npz"))
    T2 = lambda k: torch.tensor(g2[k]).half().to(DEV)
    a1 = AttentionWeights(128, None, int(g2["heads"]), DEV)
    a1.to_q.weight, a1.to_k.weight, a1.to_v.weight = T2("wq"), T2("wk"), T2("wv")
    a1.to_out[0].weight, a1.to_out[0].bias = T2("wo"), T2("bo")
    y1 = _with_attn_variant(lib, 3, lambda: AttnProcessor2_0()(a1, T2("x")))
    _close(y1, torch.tensor(g2["y"]), tol=1e-2, what="AttnProcessor2_0 (sp kernel) vs reference")


def test_region_flags_bit_exact_vs_reference_fixture(hip_lib, golden_dir):
    ops = _ops(hip_lib)
    gfile = np.load(os.path.join(golden_dir, "ip_region_masks.npz"))
    names = sorted({k[: -len("_bbox")] for k in gfile.files if k.endswith("_bbox")})
    for n in names:
        bbox = torch.tensor(gfile[n + "_bbox"])
        h, w = (int(v) for v in gfile[n + "_hw"])
        flags = ops.ip_region_flags(bbox.to(DEV), h * w, (h, w)).cpu()
        masked = torch.tensor(gfile[n + "_masked"])             # [B,N,80]: 16 dummy + 4x16
        inside_ref = masked[:, :, 16::16] == 0                   # [B,N,4]
        inside = torch.stack([(flags >> k) & 1 for k in range(4)], -1).bool()
        assert torch.equal(inside, inside_ref), n
        assert torch.equal(masked[:, :, 0] == 1, inside.any(-1)), n


def _ip_attn_ref(q, enc, bbox, hw, wk, wv, wki, wvi, heads, scale):
    from oracle.attention_ref import _heads, ip_region_mask, sdpa
    b, n, c = q.shape
    txt, ip = enc[:, :77], enc[:, 77:]
    qh = _heads(q.float(), heads)
    h_ = lambda t: _heads(t.half().float(), heads)
    t_out = sdpa(qh, h_(txt.float() @ wk.float().t()), h_(txt.float() @ wv.float().t()))
    m = ip_region_mask(bbox, n, heads, hw[0] / hw[1], 64, 16)
    i_out = sdpa(qh, h_(ip.float() @ wki.float().t()), h_(ip.float() @ wvi.float().t()), m)
    o = t_out + scale * i_out
    return o.transpose(1, 2).reshape(b, n, c)


@pytest.mark.parametrize("B,heads,hw", [(2, 2, (16, 16)), (2, 10, (32, 32)), (1, 4, (24, 40)), (1, 20, (32, 32)),
                                        (2, 10, (64, 64))])
def test_masked_ip_attention_core(hip_lib, B, heads, hw):
    from diffsensei_amd.attention_processor import LP
    ops = _ops(hip_lib)
    g = torch.Generator().manual_seed(B + heads + hw[0])
    N, C, X = hw[0] * hw[1], heads * 64, 128
    q, enc = _r((B, N, C), g), _r((B, 157, X), g)
    wk, wv, wki, wvi = (_r((C, X), g, 1 / math.sqrt(X)) for _ in range(4))
    bbox = torch.zeros(B, 4, 4)
    bbox[-1, 0] = torch.tensor([0.05, 0.10, 0.50, 0.95])
    bbox[-1, 1] = torch.tensor([0.50, 0.10, 0.95, 0.95])
    bbox[-1, 2] = torch.tensor([0.30, 0.30, 0.70, 0.60])
    ref = _ip_attn_ref(q, enc, bbox, hw, wk, wv, wki, wvi, heads, 0.6)
    encd = enc.to(DEV)
    txt, ip = ops.pad_rows(encd, 0, 77, LP), ops.pad_rows(encd, 77, 80, LP)
    kt = ops.gemm(txt.view(-1, X), wk.to(DEV)).view(B, LP, C)
    ki = ops.gemm(ip.view(-1, X), wki.to(DEV)).view(B, LP, C)
    vtt, vti = ops.gemm_batched_nt(wv.to(DEV), txt), ops.gemm_batched_nt(wvi.to(DEV), ip)
    y = ops.masked_ip_attention(q.to(DEV), kt, vtt, ki, vti, bbox.to(DEV), heads, hw, 0.6)
    _close(y, ref, tol=4e-3, what="masked ip attn")


@pytest.mark.parametrize("B,heads,hw", [(2, 2, (16, 16)), (3, 10, (32, 32)), (1, 20, (32, 64)), (2, 5, (48, 48)), (16, 20, (32, 32))])
def test_masked_ip_attention_ring_variant(hip_lib, B, heads, hw):
    """`ip_attn_kernel<8, true>` (round 4: Q tiles by LDS-DMA as whole rows into a wave-private ring of three slots, two tiles
    ahead, counted vmcnt waits; O leaves as whole rows through the same slot) forced on small and large grids - one, two,
    several tiles per block, last block with fewer tiles - must reproduce the register-staged kernel BIT FOR BIT (same
    arithmetic in the same order) and the fp32 oracle at the usual tolerance; 20 back-to-back launches keep giving the same bits
    (a counted wait that under-waits shows up as a rare wrong tile); N % 256 != 0 falls back to the register-staged kernel."""
    from diffsensei_amd import _lib
    from diffsensei_amd.attention_processor import LP
    ops = _ops(hip_lib)
    lib = _lib.load()
    g = torch.Generator().manual_seed(B * 7 + heads + hw[0])
    N, C, X = hw[0] * hw[1], heads * 64, 128
    q, enc = _r((B, N, C), g), _r((B, 157, X), g)
    wk, wv, wki, wvi = (_r((C, X), g, 1 / math.sqrt(X)) for _ in range(4))
    bbox = torch.zeros(B, 4, 4)
    bbox[-1, 0] = torch.tensor([0.05, 0.10, 0.50, 0.95])
    bbox[-1, 1] = torch.tensor([0.50, 0.10, 0.95, 0.95])
    bbox[B // 2, 2] = torch.tensor([0.30, 0.30, 0.70, 0.60])
    encd = enc.to(DEV)
    txt, ip = ops.pad_rows(encd, 0, 77, LP), ops.pad_rows(encd, 77, 80, LP)
    kt = ops.gemm(txt.view(-1, X), wk.to(DEV)).view(B, LP, C)
    ki = ops.gemm(ip.view(-1, X), wki.to(DEV)).view(B, LP, C)
    vtt, vti = ops.gemm_batched_nt(wv.to(DEV), txt), ops.gemm_batched_nt(wvi.to(DEV), ip)
    qd, bd = q.to(DEV), bbox.to(DEV)
    run = lambda: ops.masked_ip_attention(qd, kt, vtt, ki, vti, bd, heads, hw, 0.6)
    try:
        lib.ds_set_option(b"ip_attn_variant", 1)
        y1 = run().clone()
        # (round 6) variants 0 / 1 never issue the work of the padding keys 80..95 (77 text / 80 image keys: probability exactly 0);
        # variant 3 is the same kernel computing all 96 key slots - the same bits
        lib.ds_set_option(b"ip_attn_variant", 3)
        assert torch.equal(run(), y1), "skipping the padding keys 80..95 changed the result"
        for min_blocks in (1, 64, 1 << 20):          # 8, some, one tile(s) per block
            lib.ds_set_option(b"ip_attn_min_blocks", min_blocks)
            lib.ds_set_option(b"ip_attn_variant", 2)
            for _ in range(20 if min_blocks == 1 else 2):
                assert torch.equal(run(), y1), f"ring variant differs from the register-staged kernel (min_blocks {min_blocks})"
    finally:
        lib.ds_set_option(b"ip_attn_variant", 0)
        lib.ds_set_option(b"ip_attn_min_blocks", 1024)
    if B <= 3:
        _close(y1, _ip_attn_ref(q, enc, bbox, hw, wk, wv, wki, wvi, heads, 0.6), tol=4e-3, what="masked ip attn (ring)")


def test_processors_vs_reference_fixtures(hip_lib, golden_dir):
    """The HIP processors, called with the reference's processor protocol, against outputs of the reference's own
    MaskedIPAttnProcessor2_0 / AttnProcessor2_0 (fp32 on CPU): fp16 tolerance 1e-2 relative to max|y|."""
    from diffsensei_amd.attention_processor import AttentionWeights, AttnProcessor2_0, MaskedIPAttnProcessor2_0
    gfile = np.load(os.path.join(golden_dir, "masked_ip_attn.npz"))
    T = lambda k: torch.tensor(gfile[k]).half().to(DEV)
    heads = int(gfile["heads"])
    attn = AttentionWeights(128, 64, heads, DEV)
    attn.to_q.weight, attn.to_k.weight, attn.to_v.weight = T("wq"), T("wk"), T("wv")
    attn.to_out[0].weight, attn.to_out[0].bias = T("wo"), T("bo")
    proc = MaskedIPAttnProcessor2_0(128, 64, scale=float(gfile["scale"]), num_ip_tokens=64, num_dummy_tokens=16, device=DEV)
    proc.to_k_ip.weight, proc.to_v_ip.weight = T("wk_ip"), T("wv_ip")
    h, w = (int(v) for v in gfile["hw"])
    y = proc(attn, T("x"), encoder_hidden_states=T("enc"), bbox=torch.tensor(gfile["bbox"]).to(DEV), aspect_ratio=h / w)
    _close(y, torch.tensor(gfile["y"]), tol=1e-2, what="MaskedIPAttnProcessor2_0 vs reference")
    g2 = np.load(os.path.join(golden_dir, "self_attn.npz"))
    T2 = lambda k: torch.tensor(g2[k]).half().to(DEV)
    a1 = AttentionWeights(128, None, int(g2["heads"]), DEV)
    a1.to_q.weight, a1.to_k.weight, a1.to_v.weight = T2("wq"), T2("wk"), T2("wv")
    a1.to_out[0].weight, a1.to_out[0].bias = T2("wo"), T2("bo")
    y1 = AttnProcessor2_0()(a1, T2("x"))
    _close(y1, torch.tensor(g2["y"]), tol=1e-2, what="AttnProcessor2_0 vs reference")


def test_resampler_vs_reference_fixture(hip_lib, golden_dir):
    from diffsensei_amd.resampler import Resampler
    gfile = np.load(os.path.join(golden_dir, "resampler.npz"))
    sd = {k[3:]: torch.tensor(gfile[k]) for k in gfile.files if k.startswith("sd.")}
    rs = Resampler(dim=128, depth=2, dim_head=64, heads=2, num_queries=16, num_dummy_tokens=16, embedding_dim=96,
                   magi_embedding_dim=64, output_dim=256, ff_mult=4, device=DEV).load_state_dict(sd)
    y = rs(torch.tensor(gfile["in_x"]), torch.tensor(gfile["in_magi"]))
    _close(y, torch.tensor(gfile["out"]), tol=1e-2, what="Resampler vs reference")
    x0 = torch.tensor(gfile["in_x"])
    y0 = rs(torch.zeros_like(x0), torch.zeros(1, 4, 64))
    _close(y0, torch.tensor(gfile["out_zero"]), tol=1e-2, what="Resampler(zeros) vs reference")


def test_small_attention(hip_lib):
    ops = _ops(hip_lib)
    g = torch.Generator().manual_seed(21)
    for (B, heads, Nq, Nk, D) in [(2, 4, 16, 274, 64), (2, 16, 257, 257, 80), (1, 12, 197, 197, 64)]:
        q, k, v = _r((B, Nq, heads * D), g), _r((B, Nk, heads * D), g), _r((B, Nk, heads * D), g)
        hs = lambda t, n: t.float().view(B, n, heads, D).transpose(1, 2)
        ref = F.scaled_dot_product_attention(hs(q, Nq), hs(k, Nk), hs(v, Nk)).transpose(1, 2).reshape(B, Nq, heads * D)
        y = ops.small_attention(q.to(DEV), k.to(DEV), v.to(DEV), heads, D ** -0.5)
        _close(y, ref, tol=3e-3, what="small attn")


# ------------------------------------------------------------------------------------------------ embeddings / sampler
def test_time_embedding_chain(hip_lib):
    from oracle.unet_ref import timestep_sinusoid
    ops = _ops(hip_lib)
    g = torch.Generator().manual_seed(31)
    B = 4
    table = torch.zeros(2, 8)
    table[0, 0], table[1, 0] = 981.0, 41.0
    for row, t in ((0, 981.0), (1, 41.0)):
        ctr = torch.tensor([row], dtype=torch.int32, device=DEV)
        e = ops.timestep_embed(table.to(DEV), B, 320, ctr=ctr)
        ref = timestep_sinusoid(torch.full((B,), t), 320)
        _close(e, ref, tol=2e-3, what="sinusoid")
    te, tid = _r((B, 1280), g), torch.tensor([[1024, 768, 0, 0, 1024, 768]] * B).half()
    a = ops.add_time_ids(te.to(DEV), tid.to(DEV), 256)
    ref = torch.cat([te.float(), timestep_sinusoid(tid.float().flatten(), 256).reshape(B, -1)], -1)
    _close(a, ref, tol=2e-3, what="add_time_ids")
    x, w, b, add = _r((B, 1280), g), _r((512, 1280), g, 1 / 36), _r((512,), g), _r((B, 512), g)
    ref = (F.silu(x.float()).half().float() @ w.float().t() + b.float()).half().float() + add.float()
    y = ops.skinny_linear(x.to(DEV), w.to(DEV), b.to(DEV), add.to(DEV), silu_in=True)
    _close(y, ref, what="skinny linear")
    ref2 = F.silu((x.float() @ w.float().t() + b.float()).half().float())
    _close(ops.skinny_linear(x.to(DEV), w.to(DEV), b.to(DEV), silu_out=True), ref2, what="skinny silu_out")


@pytest.mark.parametrize("kind", ["euler", "ddim"])
def test_cfg_sampler_step(hip_lib, kind):
    from diffsensei_amd.schedulers import DDIMScheduler, EulerDiscreteScheduler
    from oracle.scheduler_ref import DDIMOracle, EulerDiscreteOracle
    ops = _ops(hip_lib)
    g = torch.Generator().manual_seed(41)
    ns, H, W, n = 2, 8, 12, 10
    sch = EulerDiscreteScheduler() if kind == "euler" else DDIMScheduler()
    orc = (EulerDiscreteOracle() if kind == "euler" else DDIMOracle()).set_timesteps(n)
    sch.set_timesteps(n)
    table = torch.from_numpy(sch.coef_table(7.5)).to(DEV)
    lat = (_r((ns, 4, H, W), g) * 3).contiguous()
    eps = _r((2 * ns, 4, H, W), g)
    hq = lambda t: t.half().float()
    for i in (0, 4, n - 1):
        u, c = eps.float().chunk(2)
        e = hq(u + hq(7.5 * hq(c - u)))
        ref = hq(orc.step(e, i, lat.float()))
        ctr = torch.tensor([i], dtype=torch.int32, device=DEV)
        lat_d = lat.to(DEV).clone()
        xin = torch.empty(2 * ns, H * W, 4, dtype=torch.float16, device=DEV)
        eps_nhwc = eps.permute(0, 2, 3, 1).reshape(2 * ns, H * W, 4).contiguous().to(DEV)
        ops.cfg_sampler_step(eps_nhwc, lat_d, xin, table, sch.kind, True, ctr)
        _close(lat_d, ref, tol=1.5e-3, what=f"{kind} step {i}")
        if i + 1 < n:
            nxt = hq(orc.scale_model_input(ref, i + 1))
            got = xin.view(2 * ns, H, W, 4).permute(0, 3, 1, 2)
            _close(got[:ns], nxt, tol=1.5e-3, what="next model input")
            assert torch.equal(got[:ns], got[ns:])
    # stand-alone scheduler protocol
    sch.set_timesteps(n)
    t0 = sch.timesteps[0]
    xs = sch.scale_model_input(lat.to(DEV), t0)
    _close(xs, hq(orc.scale_model_input(lat.float(), 0)), tol=1.5e-3, what="scale_model_input")
    e1 = eps[:ns]
    out = sch.step(e1.to(DEV), t0, lat.to(DEV), return_dict=False)[0]
    _close(out, hq(orc.step(e1.float(), 0, lat.float())), tol=1.5e-3, what="scheduler.step")


def test_layout_helpers(hip_lib):
    ops = _ops(hip_lib)
    g = torch.Generator().manual_seed(51)
    x = _r((3, 100, 4), g).to(DEV)
    y = ops.nhwc_to_nchw(x)
    assert torch.equal(y, x.permute(0, 2, 1).contiguous())
    assert torch.equal(ops.nchw_to_nhwc(y), x)
    e = _r((2, 157, 64), g).to(DEV)
    p = ops.pad_rows(e, 77, 80, 96)
    assert torch.equal(p[:, :80], e[:, 77:]) and p[:, 80:].abs().sum() == 0


def test_image_to_u8_bit_exact_vs_numpy(hip_lib):
    """`numpy_to_pil`'s (images * 255).round().astype("uint8") [3P] on the device, incl. exact .5 ties (half to even)."""
    ops = _ops(hip_lib)
    g = torch.Generator().manual_seed(2)
    img = torch.rand((3, 3, 40, 52), generator=g)
    img[0, 0, 0, :20] = (torch.arange(20, dtype=torch.float32) + 0.5) / 255.0      # .5 ties
    img[1, 1, 3, :4] = torch.tensor([0.0, 1.0, 0.5, 0.25])
    ref = (img.permute(0, 2, 3, 1).numpy() * 255).round().astype("uint8")
    got = ops.image_to_u8(img.to(DEV)).cpu().numpy()
    assert got.shape == ref.shape and (got == ref).all()


def test_error_reporting(hip_lib):
    from diffsensei_amd import _lib
    ops = _ops(hip_lib)
    x = torch.zeros(8, 60, dtype=torch.float16, device=DEV)
    with pytest.raises(_lib.DiffSenseiHipError, match="multiples of 8"):
        ops.gemm(x, torch.zeros(16, 60, dtype=torch.float16, device=DEV))
