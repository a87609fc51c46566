#!/usr/bin/env python
"""fp16 vs fp8 (e4m3, v_mfma_f32_32x32x64_f8f6f4) flash self-attention on the UNet's shapes at 1024^2 and 2048^2:
kernel-only time (K / V^T already quantized), the two quantisation launches, and the accuracy of both against fp32 SDPA
on a slice.  Interleaved in one process, HIP events, 10 launches per number."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from diffsensei_amd import _lib, ops  # noqa: E402
from diffsensei_amd._lib import check  # noqa: E402

lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
R = lambda *s: torch.randn(*s, generator=g, device="cuda").half()


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps * 1e3


for (B, h, N) in [(32, 20, 1024), (32, 10, 4096), (2, 20, 4096), (2, 10, 16384), (8, 10, 16384)]:
    C = h * 64
    q, k, vt = R(B, N, C), R(B, N, C), R(B, h, 64, N)
    st = torch.cuda.current_stream().cuda_stream
    o16 = ops.self_attention(q, k, vt, h)
    k8 = ops.quantize_fp8(k)
    v8 = ops.quantize_fp8(vt.reshape(B * h, 64, N), permute64=True)
    o8 = torch.empty_like(q)
    run8 = lambda: check(lib.ds_self_attn_fp8_f16(q.data_ptr(), C, N * C, k8.data_ptr(), v8.data_ptr(), o8.data_ptr(), C, N * C,
                                                   B, h, N, N, 0.125, st))
    t16 = timed(lambda: ops.self_attention(q, k, vt, h))
    t8 = timed(run8)
    tq = timed(lambda: (ops.quantize_fp8(k), ops.quantize_fp8(vt.reshape(B * h, 64, N), permute64=True)))
    fl = 4.0 * B * h * N * N * 64
    nq = min(N, 512)    # accuracy on the first head of the first image, first 512 queries
    qs, ks, vs = q[0, :nq, :64].float(), k[0, :, :64].float(), vt[0, 0].float().t()
    ref = F.scaled_dot_product_attention(qs[None], ks[None], vs[None])[0]
    rel = lambda o: ((o[0, :nq, :64].float() - ref).norm() / ref.norm()).item()
    print(f"B={B:2d} h={h:2d} N={N:5d} | f16 {t16:8.1f} us {fl / t16 / 1e6:6.0f} TF | fp8 {t8:8.1f} us {fl / t8 / 1e6:6.0f} TF "
          f"(+quant {tq:6.1f} us) | speedup {t16 / t8:.2f}x ({t16 / (t8 + tq):.2f}x with quant) | rel-L2 f16 {rel(o16):.2e} fp8 {rel(o8):.2e}",
          flush=True)
