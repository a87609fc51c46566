#!/usr/bin/env python
"""Run ONE GEMM shape/variant a few times (target for rocprofv3 --pmc passes).

    python tools/one_gemm.py M N K variant [reps] [geglu]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from diffsensei_amd import _lib, ops  # noqa: E402

M, N, K, variant = (int(v) for v in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
geglu = len(sys.argv) > 6 and sys.argv[6] in ("geglu", "geglu_ln")
fused_ln = len(sys.argv) > 6 and sys.argv[6].endswith("_ln")     # the fused-LayerNorm consumer instantiation (round 4)
lib = _lib.load()
lib.ds_set_option(b"gemm_variant", variant)
g = torch.Generator(device="cuda").manual_seed(0)
x = (torch.randn(M, K, generator=g, device="cuda") * 0.5).half()
w = (torch.randn(N, K, generator=g, device="cuda") * K ** -0.5).half()
b = torch.randn(N, generator=g, device="cuda").half()
if fused_ln:
    from diffsensei_amd.engine import pack_geglu, pack_ln_fused
    gamma, beta = (1 + 0.1 * torch.randn(K, generator=g, device="cuda")).half(), (0.1 * torch.randn(K, generator=g, device="cuda")).half()
    gw, c2, b2 = pack_ln_fused(w, b, gamma, beta)
    if geglu:
        gw, b2 = pack_geglu(gw, b2)
        half = c2.shape[0] // 2
        c2 = torch.stack([c2[:half].reshape(-1, 64, 2), c2[half:].reshape(-1, 64, 2)], dim=1).reshape(-1, 2).contiguous()
    xs = x.float().view(M, K // 64, 64)
    st = ops.ln_finalize(torch.stack([xs.sum(-1).t(), (xs * xs).sum(-1).t()], dim=-1).contiguous(), K, 1e-5)
    del xs
    run = lambda out=None: ops.gemm_ln(x, gw, b2, c2, st, geglu=geglu, out=out)
else:
    run = lambda out=None: ops.gemm(x, w, b, geglu=geglu, out=out)
y = run()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(reps):
    run(y)
ev[1].record()
torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / reps
print(f"M={M} N={N} K={K} variant={variant}: {ms * 1e3:.1f} us, {2.0 * M * N * K / ms / 1e9:.1f} TF/s")
