#!/usr/bin/env python
"""Run ONE operator a few times (target for rocprofv3 --pmc passes): the bench's second and third kernels.

    python tools/one_op.py conv   [reps]     conv_halo256_kernel: B=32, 32x32, 1280 -> 1280 channels (a level-2 ResNet conv)
    python tools/one_op.py attn   [reps]     self_attn_kernel<2>: B=32, 10 heads, N=4096 (level-1 self-attention)
    python tools/one_op.py attn1k [reps]     self_attn_kernel<1>: B=32, 20 heads, N=1024 (level-2 self-attention)
    python tools/one_op.py conv_b2 [reps]    conv_halo_deep_kernel: B=2, 32x32, 1280 -> 1280 (the level-2 conv of a batch-1 request)
    python tools/one_op.py t160 [reps]       gemm_t160_kernel: M=2048, N=1280, K=1280 + bias + residual
    python tools/one_op.py t160tall [reps]   gemm_t160_kernel<4,4> (128 x 160 tiles): M=2048, N=2560, K=1280 + bias (q|k of a batch-1 request)
    python tools/one_op.py g320 [reps]       gemm_g320_kernel: M=2048, N=10240 packed, K=1280 + bias, GEGLU (a batch-1 request's FF projection)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from diffsensei_amd import _lib, ops  # noqa: E402

what = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
_lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
R = lambda *s: (torch.randn(*s, generator=g, device="cuda") * 0.5).half()
if what in ("conv", "conv_b2"):
    B, H, C = (32 if what == "conv" else 2), 32, 1280
    x, w, b = R(B, H, H, C), R(C, 3, 3, C) * ((9 * C) ** -0.5) * 2, R(C)
    fn = lambda: ops.conv3x3(x, w, b)
    flop = 2.0 * B * H * H * C * C * 9
elif what == "t160":
    M, N, K = 2048, 1280, 1280
    x, w, b, r = R(M, K), R(N, K) * (K ** -0.5), R(N), R(M, N)
    fn = lambda: ops.gemm(x, w, b, r)
    flop = 2.0 * M * N * K
elif what == "t160tall":
    M, N, K = 2048, 2560, 1280
    x, w, b = R(M, K), R(N, K) * (K ** -0.5), R(N)
    fn = lambda: ops.gemm(x, w, b)
    flop = 2.0 * M * N * K
elif what == "g320":
    from diffsensei_amd.engine import pack_geglu320
    M, N, K = 2048, 10240, 1280
    x, w, b = R(M, K), pack_geglu320(R(N, K) * (K ** -0.5)), pack_geglu320(R(N))
    fn = lambda: ops.gemm(x, w, b, geglu=320)
    flop = 2.0 * M * N * K
else:
    B, heads, N = (32, 10, 4096) if what == "attn" else (32, 20, 1024)
    C = heads * 64
    q, k, vt = R(B, N, C), R(B, N, C), R(B, heads, 64, N)
    fn = lambda: ops.self_attention(q, k, vt, heads)
    flop = 4.0 * B * heads * N * N * 64
fn()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(reps):
    fn()
ev[1].record()
torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / reps
print(f"{what}: {ms * 1e3:.1f} us, {flop / ms / 1e9:.1f} TF/s")
