#!/usr/bin/env python
"""gemm_g320_kernel (256 x 320 tiles, one block per CU) against the 128-row-packed GEGLU kernels on the GEGLU projection of a
small-batch request, back to back, interleaved rounds, HIP events:   python tools/one_g320.py [M N K] [reps]
Both the plain (bias) form and the fused-LayerNorm consumer form (partial sums of a producer launch)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsensei_amd import _lib, ops
from diffsensei_amd.engine import pack_geglu, pack_geglu320, pack_ln_fused

a = [int(v) for v in sys.argv[1:]]
M, N, K = (a + [2048, 10240, 1280])[:3] if len(a) >= 3 else (2048, 10240, 1280)
reps = a[3] if len(a) > 3 else 20
lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
x = (torch.randn(M, K, generator=g, device="cuda") * 0.5).half()
w = (torch.randn(N, K, generator=g, device="cuda") * K ** -0.5).half()
b = torch.randn(N, generator=g, device="cuda").half()
gamma, beta = (1 + 0.1 * torch.randn(K, generator=g, device="cuda")).half(), (0.1 * torch.randn(K, generator=g, device="cuda")).half()
gw, c2, b2 = pack_ln_fused(w, b, gamma, beta)
half = N // 2
wp, bp = pack_geglu(w, b)
gwp, b2p = pack_geglu(gw, b2)
c2p = torch.stack([c2[:half].reshape(-1, 64, 2), c2[half:].reshape(-1, 64, 2)], dim=1).reshape(-1, 2).contiguous()
w3, b3, gw3, b23, c23 = pack_geglu320(w), pack_geglu320(b), pack_geglu320(gw), pack_geglu320(b2), pack_geglu320(c2)
xs = x.float().view(M, K // 64, 64)
part = torch.stack([xs.sum(-1).t(), (xs * xs).sum(-1).t()], dim=-1).contiguous()
del xs
y = torch.empty((M, half), dtype=torch.float16, device="cuda")
runs = {
    "plain  128-packed (auto dispatch)": lambda: ops.gemm(x, wp, bp, geglu=True, out=y),
    "plain  gemm_g320_kernel          ": lambda: ops.gemm(x, w3, b3, geglu=320, out=y),
    "LN     128-packed consumer       ": lambda: ops.gemm_ln_partial(x, gwp, b2p, c2p, part, geglu=True, out=y),
    "LN     gemm_g320_kernel consumer ": lambda: ops.gemm_ln_partial(x, gw3, b23, c23, part, geglu=320, out=y),
}
outs, t = {}, {k: [] for k in runs}
for rnd in range(5):
    for k, f in runs.items():
        f()
        torch.cuda.synchronize()
        outs[k] = y.clone()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(reps):
            f()
        ev[1].record()
        torch.cuda.synchronize()
        t[k].append(ev[0].elapsed_time(ev[1]) / reps * 1e3)
ks = list(runs)
print(f"M={M} N={N} K={K}: plain equal {torch.equal(outs[ks[0]], outs[ks[1]])}, LN equal {torch.equal(outs[ks[2]], outs[ks[3]])}")
for k in ks:
    print(f"  {k}: min {min(t[k]):7.1f} us ({2.0 * M * N * K / min(t[k]) / 1e6:6.1f} TF/s)  med {statistics.median(t[k]):7.1f}")
