#!/usr/bin/env python
"""End-to-end A/B of the halo-conv block shape (0 = library's choice, 1 = 8x16 px, 2 = 16x16 px) on small-batch
resolution buckets: denoise-loop ms per step.  The launch plan is rebuilt per variant (the choice is baked at capture)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from diffsensei_amd import _lib
dev = torch.device("cuda", 0)
pipe, _ = bench.build_pipeline(dev, 1, 0)
lib = _lib.load()
for size, ns, steps in [(1536, 2, 6), (2048, 1, 4), (1024, 4, 10)]:
    req = bench.synthetic_request(dev, size, seed=size)
    req["num_inference_steps"] = steps
    req["output_type"] = "latent"
    for variant in (0, 1, 2, 0):
        assert lib.ds_set_option(b"conv_halo_variant", variant) == 0
        pipe.unet._engines.clear()
        pipe(num_samples=ns, **req)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pipe(num_samples=ns, **req)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print(json.dumps({"size": size, "num_samples": ns, "conv_halo_variant": variant,
                          "ms_per_denoise_step": round(best / steps * 1e3, 2)}), flush=True)
lib.ds_set_option(b"conv_halo_variant", 0)
