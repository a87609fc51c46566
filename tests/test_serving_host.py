"""CPU: the request front-end's batching plan (diffsensei_amd/serving.py) with a recording stand-in for the pipeline."""
import pytest

from diffsensei_amd.serving import BucketBatcher, bucket_key, plan_batches


def _req(size, ns=1, steps=50, g=7.5, **kw):
    return dict(prompt="p", height=size, width=size, num_inference_steps=steps, guidance_scale=g, num_samples=ns, **kw)


def test_plan_batches_buckets_and_caps():
    reqs = [_req(512), _req(1024, 2), _req(512, 3), _req(1024), _req(768), _req(512), _req(1024, 2), _req(512, 1, g=5.0)]
    plan = plan_batches(reqs, max_panels=4)
    flat = sorted(i for b in plan for i in b)
    assert flat == list(range(len(reqs)))                                    # every request exactly once
    for b in plan:
        assert len({bucket_key(reqs[i]) for i in b}) == 1                    # never mixes buckets
        assert sum(reqs[i]["num_samples"] for i in b) <= 4                   # panel cap
    assert [reqs[b[0]]["height"] for b in plan] == sorted((reqs[b[0]]["height"] for b in plan), reverse=True)
    assert plan[0] == [1, 3] and plan[1] == [6]                              # 1024: 2+1 panels, then 2 (cap 4)
    assert [0, 2] in plan and [5] in plan and [7] in plan                    # 512: order kept; guidance 5.0 alone
    with pytest.raises(ValueError):
        plan_batches([_req(512, 8)], max_panels=4)


def test_pixel_cap_shapes_packing_not_single_requests():
    """with the 32 Mpx default a 2048 x 2048 request of 9..16 samples is above the pixel cap (8 panels); it must
    still be served - alone - as it was before the cap existed; only `num_samples > max_panels` is an error."""
    reqs = [_req(2048, 12), _req(2048, 2), _req(2048, 5), _req(2048, 2), _req(512, 20)]
    plan = plan_batches(reqs, max_panels=32, max_pixels=32 * 1024 * 1024)
    assert plan[:3] == [[0], [1, 2], [3]] and plan[3] == [4]      # 12 alone; 2 + 5 <= 8 packed; 2 after the flush; 512: cap 32
    with pytest.raises(ValueError):
        plan_batches([_req(2048, 33)], max_panels=32, max_pixels=32 * 1024 * 1024)


def test_batcher_routes_results_by_ticket():
    calls = []

    class FakePipe:
        def generate_batch(self, requests, output_type="pil"):
            calls.append(([r["tag"] for r in requests], output_type))
            return [f"{r['tag']}:{output_type}" for r in requests]

    def req(size, tag, ns=1):
        return dict(prompt="p", height=size, width=size, num_inference_steps=50, guidance_scale=7.5, num_samples=ns, tag=tag)

    b = BucketBatcher(FakePipe(), max_panels=2)
    tickets = [b.submit(**req(512, "a")), b.submit(**req(1024, "b")), b.submit(**req(512, "c")), b.submit(**req(512, "d"))]
    assert tickets == [0, 1, 2, 3] and len(b) == 4
    with pytest.raises(TypeError):
        b.submit(output_type="pt", **req(512, "x"))
    out = b.run(output_type="pt")
    assert out == ["a:pt", "b:pt", "c:pt", "d:pt"] and len(b) == 0
    assert calls == [(["b"], "pt"), (["a", "c"], "pt"), (["d"], "pt")]
