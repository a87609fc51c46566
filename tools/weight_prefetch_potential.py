#!/usr/bin/env python
"""What would a num_samples-1 GEMM gain if its weight were already in the Infinity Cache when it starts?

At UNet batch 2 every weight is read ONCE per forward, cold from HBM (5.8 GB of weights against 256 MiB of Infinity Cache), and
the small-grid GEMMs run 33 us where the same launch with a cache-hot weight runs 18...24 us (profiles/r02_small_batch_gemm_ab.txt,
r02_ring_in_pipeline_ab.txt).  This tool prices a weight prefetch before any kernel is changed:

    hot       the same weight every launch
    cold      a pool of weights larger than the Infinity Cache, used round-robin (what the sampler sees)
    ahead     cold, but a side stream reads one element per 128-byte line of weight i+1 while GEMM i runs (paced by events,
              so the prefetch is never more than one weight ahead)
    touched   cold, the touch of weight i runs on the SAME stream right before GEMM i (upper bound: the weight is as warm as a
              prefetch can make it; only the GEMMs are timed)

    python tools/weight_prefetch_potential.py [launches]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from diffsensei_amd import build, ops  # noqa: E402

assert open(os.path.join(build.LIBDIR, "build.stamp")).read().strip() == build._digest(), "sources changed after the library was built"
n_launch = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
SHAPES = [("proj_L2", 2048, 1280, 1280), ("ff2_L2", 2048, 1280, 5120), ("ff1_L2", 2048, 10240, 1280), ("qkv_L1", 8192, 640, 640),
          ("ff2_L1", 8192, 640, 2560)]
POOL_BYTES = 640 << 20


def touch(w):
    return w.view(-1, 64)[:, 0].sum()


for name, M, N, K in SHAPES:
    wbytes = N * K * 2
    pool = max(4, POOL_BYTES // wbytes)
    ws = [(torch.randn(N, K, generator=g, device=dev) * K ** -0.5).half() for _ in range(pool)]
    x = (torch.randn(M, K, generator=g, device=dev) * 0.5).half()
    b = torch.randn(N, generator=g, device=dev).half()
    y = torch.empty(M, N, dtype=torch.float16, device=dev)
    s1 = torch.cuda.current_stream()
    s2 = torch.cuda.Stream()
    res = {}
    for mode in ("hot", "cold", "ahead", "touched", "cold", "ahead"):
        for rep in range(2):   # first pass = warm-up of the mode
            torch.cuda.synchronize()
            starts = [torch.cuda.Event(enable_timing=True) for _ in range(n_launch)]
            ends = [torch.cuda.Event(enable_timing=True) for _ in range(n_launch)]
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for i in range(n_launch):
                w = ws[0] if mode == "hot" else ws[i % pool]
                if mode == "ahead":
                    if i > 0:
                        s2.wait_event(ends[i - 1])
                    with torch.cuda.stream(s2):
                        touch(ws[(i + 1) % pool])
                elif mode == "touched":
                    touch(w)
                starts[i].record()
                ops.gemm(x, w, b, out=y)
                ends[i].record()
            t1.record()
            torch.cuda.synchronize()
        per = sorted(starts[i].elapsed_time(ends[i]) for i in range(8, n_launch))
        res.setdefault(mode, []).append((per[len(per) // 2] * 1e3, t0.elapsed_time(t1) / n_launch * 1e3))
    line = f"{name:8s} M={M:5d} N={N:5d} K={K:5d} pool {pool:3d} x {wbytes / 2**20:5.1f} MiB |"
    for mode in ("hot", "cold", "ahead", "touched"):
        line += f" {mode} " + " ".join(f"{m:6.1f}us (loop {t:6.1f})" for m, t in res[mode]) + " |"
    print(line, flush=True)
    del ws
    torch.cuda.empty_cache()
