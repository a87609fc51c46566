"""Request front-end: resolution-bucketed batching of panel requests.

The reference serves one panel request at a time from a Gradio callback (scripts/demo/gradio_wo_mllm.py:45, the
`pipeline(...)` call inside `result_generation`), and its training side groups images by size with a bucket sampler
(src/datasets/dataset_size_bucket.py:488-544: items are binned by (height, width) and a batch never mixes bins).
Here the same idea is applied to inference, because one UNet launch plan / hipGraph exists per
(batch, height, width) and the kernels only reach their throughput on large batches:

    batcher = BucketBatcher(pipe)                              # up to 32 panels per UNet batch at 1024 x 1024
    tickets = [batcher.submit(**request_kwargs) for ...]      # the keyword arguments of DiffSenseiPipeline.__call__
    results = batcher.run(output_type="pil")                   # results[ticket] = that request's images

Requests that share (height, width, steps, guidance_scale, ip_scale) are concatenated into one UNet batch of at most
`max_panels` panels (`DiffSenseiPipeline.generate_batch`: per-request prompts, character references, boxes, seeds);
buckets run largest-resolution first.  Multi-GPU: shard the request list with `distributed.shard_requests` (LPT by
pixel count, no data-path collective) and run one batcher per rank.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Tuple


def bucket_key(request: dict) -> Tuple:
    """What must agree for two requests to share a UNet batch."""
    return (request.get("height"), request.get("width"), request.get("num_inference_steps", 40),
            float(request.get("guidance_scale", 5.0)), float(request.get("ip_scale", 1.0)))


def plan_batches(requests: List[dict], max_panels: int, max_pixels: Optional[int] = None) -> List[List[int]]:
    """Indices of `requests` grouped into batches: same bucket, at most `max_panels` panels (sum of num_samples) per
    batch - and at most `max_pixels` output pixels if given, so small resolutions get proportionally larger batches (a
    single request above the pixel cap but within `max_panels` runs alone; only `num_samples > max_panels` is an error) -
    submission order kept inside a bucket, buckets ordered by decreasing pixel count (the long jobs first)."""
    if max_panels < 1:
        raise ValueError("max_panels must be >= 1")
    buckets: Dict[Tuple, List[int]] = {}
    for i, r in enumerate(requests):
        buckets.setdefault(bucket_key(r), []).append(i)
    order = sorted(buckets, key=lambda k: -((k[0] or 0) * (k[1] or 0)))
    batches: List[List[int]] = []
    for k in order:
        cap = max_panels
        if max_pixels is not None and k[0] and k[1]:
            cap = max(1, min(max_panels, max_pixels // (k[0] * k[1])))
        cur, panels = [], 0
        for i in buckets[k]:
            n = int(requests[i].get("num_samples", 1) or 1)
            if n > max_panels:
                raise ValueError(f"request {i}: num_samples {n} exceeds max_panels {max_panels}")
            if n > cap:
                # the pixel cap only shapes how requests are PACKED: a single request that is larger than it (2048 x 2048 with
                # num_samples 9..16 under the 32 Mpx default) is still served, in a batch of its own, as before the cap existed
                if cur:
                    batches.append(cur)
                    cur, panels = [], 0
                batches.append([i])
                continue
            if cur and panels + n > cap:
                batches.append(cur)
                cur, panels = [], 0
            cur.append(i)
            panels += n
        if cur:
            batches.append(cur)
    return batches


class BucketBatcher:
    """Collects requests, then runs them bucket by bucket through `pipe.generate_batch`."""

    def __init__(self, pipe, max_panels: int = 32, max_pixels: Optional[int] = 32 * 1024 * 1024):
        # defaults = the benchmark's operating point (bench.py: 32 panels of 1024 x 1024 per call = UNet batch 64, where every
        # projection of the level-2 transformers is a whole number of 256-tile rounds); the pixel cap scales the panel count
        # down for larger images and lets smaller ones use the full 32
        self.pipe = pipe
        self.max_panels = max_panels
        self.max_pixels = max_pixels
        self._pending: List[dict] = []
        self.last_plan: List[List[int]] = []

    def submit(self, **request) -> int:
        """Queue one request (keyword arguments of `DiffSenseiPipeline.__call__`, without `output_type`); returns its ticket."""
        if "output_type" in request:
            raise TypeError("output_type is chosen per run(), not per request")
        self._pending.append(request)
        return len(self._pending) - 1

    def __len__(self) -> int:
        return len(self._pending)

    def run(self, output_type: str = "pil") -> List[Any]:
        """Run everything queued; returns the per-request outputs indexed by ticket and empties the queue."""
        reqs, self._pending = self._pending, []
        self.last_plan = plan_batches(reqs, self.max_panels, self.max_pixels)
        results: List[Any] = [None] * len(reqs)
        for batch in self.last_plan:
            outs = self.pipe.generate_batch([reqs[i] for i in batch], output_type=output_type)
            for i, o in zip(batch, outs):
                results[i] = o
        return results
