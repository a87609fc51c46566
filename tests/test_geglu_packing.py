"""CPU: the two GEGLU weight packings of the launch planner (diffsensei_amd/engine.py) - 128-row groups (64 hidden + 64 gates:
gemm.hip / gemm_pp.hip) and 320-row groups (160 hidden + 160 gates: gemm_g320.hip) - are permutations of the reference layout
[hidden rows | gate rows] of diffusers' GEGLU.proj [3P] (reached from reference src/models/unet.py:244-338), and one converts
into the other the way PackedUNet.geglu320 does it."""
import torch

from diffsensei_amd.engine import pack_geglu, pack_geglu320, unpack_geglu


def test_geglu_packings_are_inverse_permutations():
    g = torch.Generator().manual_seed(0)
    C = 160                                  # inner width 4 C = 640: a multiple of 64 and of 160
    w, b = torch.randn(8 * C, C, generator=g), torch.randn(8 * C, generator=g)
    wp, bp = pack_geglu(w, b)
    assert torch.equal(unpack_geglu(wp), w) and torch.equal(unpack_geglu(bp), b)
    w3, b3 = pack_geglu320(w), pack_geglu320(b)
    half = 4 * C
    for t in range(half // 160):             # group t: hidden rows 160 t .., then the matching gate rows
        assert torch.equal(w3[320 * t:320 * t + 160], w[160 * t:160 * t + 160])
        assert torch.equal(w3[320 * t + 160:320 * t + 320], w[half + 160 * t:half + 160 * t + 160])
        assert torch.equal(b3[320 * t:320 * t + 160], b[160 * t:160 * t + 160])
    assert sorted(b3.tolist()) == sorted(b.tolist())
    c2 = torch.randn(8 * C, 2, generator=g)  # the fused-LayerNorm (-c hi, -c lo) pairs travel with their rows
    assert torch.equal(pack_geglu320(c2)[:, 0], pack_geglu320(c2[:, 0].contiguous()))
    # the planner's route: 128-packed -> reference order -> 320-packed
    assert torch.equal(pack_geglu320(unpack_geglu(wp)), w3)
