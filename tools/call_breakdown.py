#!/usr/bin/env python
"""Where one DiffSenseiPipeline.__call__ spends its wall time: conditioning (text encoders, CLIP-H / ViT-MAE / Resampler),
the request prologue of the denoise loop (hoisted K / V projections, schedule upload, plan lookup), the 50 plan replays, the VAE
decode + uint8 + D2H + PIL tail.  Each phase is bracketed by a device synchronisation, so the sum is an upper bound of the call.
    python tools/call_breakdown.py [num_samples] [refs] [dialog 0|1] [calls]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 1
refs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dialog = int(sys.argv[3]) if len(sys.argv) > 3 else 0
calls = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = torch.device("cuda", 0)
pipe, _ = bench.build_pipeline(dev, 1, 0, with_vae=True, keep_oracle=False)
req = bench.synthetic_request(dev, 1024, seed=1234, output_type="pil", refs=refs)
if not dialog:
    req["dialog_bbox"] = []
acc = {}


def timed(obj, name, label=None):
    fn = getattr(obj, name)
    def wrap(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); acc[label or name] = acc.get(label or name, 0.0) + time.perf_counter() - t0
        return r
    setattr(obj, name, wrap)


for n in ("_conditioning", "_denoise", "_postprocess", "encode_prompt", "prepare_ip_image_embeds", "prepare_latents"):
    timed(pipe, n)
pipe(num_samples=ns, **req)          # warm-up: plans, graph capture
eng = next(iter(pipe.unet._engines.values())) if hasattr(pipe.unet, "_engines") else None
if eng is not None:
    timed(eng, "set_request"); timed(eng, "load_schedule"); timed(eng, "build_sampler")
    timed(eng.prep_plan, "run", "prep_plan.run")
acc.clear()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(calls):
    pipe(num_samples=ns, **req)
torch.cuda.synchronize(); total = (time.perf_counter() - t0) / calls
print(f"num_samples {ns}, refs {refs}, dialog {dialog}: {total * 1e3:.1f} ms per call with the phase synchronisations "
      f"({ns / total:.4f} panels/s)")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:28s} {v / calls * 1e3:9.2f} ms")
steps = acc["_denoise"] / calls - sum(acc.get(k, 0.0) for k in ("set_request", "load_schedule", "build_sampler", "prep_plan.run")) / calls
print(f"  -> 50 plan replays (by difference) {steps * 1e3:9.2f} ms = {steps * 1e3 / 50:.3f} ms per step; "
      f"info {pipe.last_run_info}")
