#!/usr/bin/env python
"""Resolution-bucket sweep (BASELINE.json configs[3]/[4] shapes) on one GPU: denoise-loop throughput per bucket."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
pipe, _ = bench.build_pipeline(dev, 1, 0)
rows = []
for size, ns, steps in [(512, 8, 10), (768, 4, 10), (1024, 4, 10), (1536, 2, 6), (2048, 1, 4),
                        (512, 32, 6), (768, 16, 6), (1536, 8, 4), (2048, 4, 4)]:
    req = bench.synthetic_request(dev, size, seed=size)
    req["num_inference_steps"] = steps
    req["output_type"] = "latent"                    # denoise loop only (the VAE decode is timed by tools/vae_bench.py)
    pipe(num_samples=ns, **req)                      # warm-up builds + captures the plan for this bucket
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = pipe(num_samples=ns, **req).images
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert torch.isfinite(out.float()).all()
    per_step = dt / steps
    rows.append({"size": size, "num_samples": ns, "ms_per_denoise_step": round(per_step * 1e3, 2),
                 "panels_per_s_at_50_steps": round(ns / (per_step * 50), 4)})
    print(json.dumps(rows[-1]), flush=True)
