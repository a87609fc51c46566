"""GPU: GroupNorm statistics out of the producing convolution (csrc/conv_halo.hip `gn_emit`, DsOp CONV3X3 p[6] / GROUPNORM
i[6]): conv1 -> norm2 of diffusers' ResnetBlock2D [3P], reached from reference src/models/unet.py:244-338.  The convolution's
epilogue leaves one (sum, sum of squares) pair per image, pixel tile and channel in the GroupNorm workspace; the GroupNorm then
runs finalize + apply only.  Checked against (a) the three-launch GroupNorm on the same conv output (same statistics up to fp32
summation order: <= 2e-3 max-relative on the normalised output, which is what a 1e-6 relative change of a variance does after
f16 rounding) and (b) plain PyTorch fp32 GroupNorm + SiLU of the stored conv output (<= 3e-3, the tolerance of the GroupNorm
tests); the convolution's own output must be bit-identical with and without the statistics."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _r(shape, g, scale=1.0):
    return (torch.randn(shape, generator=g) * scale).half()


def _run(lib, op):
    rc = lib.ds_op_run(C.byref(op), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, lib.ds_last_error().decode()


@pytest.mark.parametrize("B,H,W,Cin,Cout,variant", [(2, 32, 32, 1280, 1280, 0), (2, 32, 32, 640, 320, 2), (3, 18, 13, 64, 320, 1),
                                                    (16, 32, 32, 320, 640, 2), (2, 64, 64, 320, 320, 0), (1, 40, 24, 128, 192, 2),
                                                    (2, 32, 32, 1280, 1280, 3), (2, 32, 32, 1280, 1280, 1)])
def test_groupnorm_from_conv_partials(hip_lib, B, H, W, Cin, Cout, variant):
    from diffsensei_amd import _lib, ops
    from diffsensei_amd.engine import make_op
    lib = _lib.load()
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + Cin + Cout)
    x = _r((B, H, W, Cin), g).to(DEV)
    w = _r((Cout, 3, 3, Cin), g, (9 * Cin) ** -0.5).to(DEV)
    b = _r((Cout,), g, 0.3).to(DEV)
    rb = _r((B, Cout), g, 0.5).to(DEV)             # the time-embedding projection (per-image bias), as conv1 carries it
    gamma, beta = (1 + 0.2 * torch.randn(Cout, generator=g)).half().to(DEV), _r((Cout,), g, 0.2).to(DEV)
    assert lib.ds_set_option(b"conv_halo_variant", variant) == 0
    try:
        nch = int(lib.ds_conv3x3_gn_chunks(B, H, W, Cin, Cout))
        assert nch > 0, "the halo-patch kernels take every stride-1 shape with Cin % 64 == 0"
        ws = torch.zeros(lib.ds_groupnorm_workspace_bytes(B, Cout), dtype=torch.uint8, device=DEV)
        y = torch.empty((B, H, W, Cout), dtype=torch.float16, device=DEV)
        _run(lib, make_op("CONV3X3", i=(B, H, W, Cin, Cout, 1, 0, Cout, 0, 0), p=(x, w, y, b, rb, None, ws)))
        plain = ops.conv3x3(x, w, b, rowbias=rb)
        assert torch.equal(y, plain), "emitting the statistics changed the convolution's output"
        # the partial sums really are those of the stored tensor: per image and channel, over all tiles
        part = ws[: B * nch * Cout * 8].view(torch.float32).view(B, nch, Cout, 2).sum(1).cpu()
        yf = y.float().cpu().view(B, H * W, Cout)
        assert torch.allclose(part[..., 0], yf.sum(1), rtol=1e-4, atol=2e-2)
        assert torch.allclose(part[..., 1], (yf * yf).sum(1), rtol=1e-4, atol=2e-2)
        out = torch.empty((B, H * W, Cout), dtype=torch.float16, device=DEV)
        _run(lib, make_op("GROUPNORM", i=(B, H * W, Cout, 0, 32, 1, nch), f=(1e-5,), p=(y, None, out, gamma, beta, ws)))
    finally:
        lib.ds_set_option(b"conv_halo_variant", 0)
    three = ops.groupnorm(y.view(B, H * W, Cout), gamma, beta, 32, 1e-5, True)
    ref = F.silu(F.group_norm(yf.transpose(1, 2), 32, gamma.float().cpu(), beta.float().cpu(), 1e-5)).transpose(1, 2)
    den = ref.abs().max().item()
    e3 = (out.float().cpu() - three.float().cpu()).abs().max().item() / den
    er = (out.float().cpu() - ref).abs().max().item() / den
    print(f"GroupNorm from conv partials B={B} {H}x{W} {Cin}->{Cout} ({nch} chunks): vs three-launch {e3:.2e}, vs fp32 {er:.2e}")
    assert torch.isfinite(out).all() and e3 <= 2e-3 and er <= 3e-3, (e3, er)


def test_gn_chunks_query_and_refusals(hip_lib):
    from diffsensei_amd import _lib
    from diffsensei_amd.engine import make_op
    lib = _lib.load()
    q = lambda *a: int(lib.ds_conv3x3_gn_chunks(*a))
    assert q(2, 128, 128, 320, 320) == 128 and q(64, 128, 128, 320, 320) == 64      # 8 x 16 tiles / 16 x 16 tiles
    assert q(2, 32, 32, 1280, 1280) == 8 and q(64, 32, 32, 1280, 1280) == 4
    assert q(2, 256, 256, 320, 320) == 0                                            # 2048 x 2048: more tiles than the workspace holds
    assert q(2, 32, 32, 4, 320) == 0                                                # conv_in's shape: not a halo-patch kernel shape
    # a GroupNorm told about more chunks than its workspace holds is refused, not mis-served
    x = torch.zeros((1, 64, 32), dtype=torch.float16, device=DEV)
    ws = torch.zeros(lib.ds_groupnorm_workspace_bytes(1, 32), dtype=torch.uint8, device=DEV)
    gb = torch.ones(32, dtype=torch.float16, device=DEV)
    op = make_op("GROUPNORM", i=(1, 64, 32, 0, 32, 0, 129), f=(1e-5,), p=(x, None, x.clone(), gb, gb, ws))
    assert lib.ds_op_run(C.byref(op), C.c_void_p(torch.cuda.current_stream().cuda_stream)) != 0
