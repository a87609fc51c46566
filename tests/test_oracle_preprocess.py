"""Pins oracle/image_preprocess_ref.py: the numpy restatement of Pillow's 8-bit separable resize must be BIT-EXACT against
Pillow itself (installed third-party code, the arithmetic behind reference src/pipelines/pipeline_diffsensei.py:125-126),
and the two processor pipelines must match transformers' CLIPImageProcessor / ViTImageProcessor (fp32, <= 1e-6: the
library fuses rescale and normalisation in a different order).  CPU only."""
import numpy as np
import pytest
from PIL import Image

from oracle import image_preprocess_ref as P

SHAPES = [((300, 200), (336, 224)), ((224, 224), (224, 224)), ((512, 640), (224, 280)), ((100, 100), (224, 224)),
          ((386, 224), (224, 224)), ((37, 911), (224, 551)), ((640, 640), (224, 224)), ((224, 312), (224, 224)),
          ((5, 7), (224, 224)), ((225, 223), (224, 224))]


@pytest.mark.parametrize("src,dst", SHAPES)
@pytest.mark.parametrize("name,flag", [("bicubic", Image.BICUBIC), ("bilinear", Image.BILINEAR)])
def test_resize_is_bit_exact_against_pillow(src, dst, name, flag):
    rng = np.random.RandomState(src[0] * 1000 + src[1])
    for kind in ("noise", "flat", "edges"):
        if kind == "noise":
            img = rng.randint(0, 256, src + (3,), dtype=np.uint8)
        elif kind == "flat":
            img = np.full(src + (3,), 255, np.uint8)                     # overshoot of the negative bicubic lobes clamps
        else:
            img = (rng.randint(0, 2, src + (3,)) * 255).astype(np.uint8)
        ref = np.asarray(Image.fromarray(img).resize((dst[1], dst[0]), flag))
        assert np.array_equal(P.pil_resize_u8(img, dst[0], dst[1], name), ref), (kind, src, dst, name)


def test_shortest_edge_rule():
    assert P.shortest_edge_size(300, 200) == (336, 224)
    assert P.shortest_edge_size(224, 386) == (224, 386)
    assert P.shortest_edge_size(97, 333) == (224, int(224 * 333 / 97))
    assert P.shortest_edge_size(500, 500) == (224, 224)


@pytest.mark.parametrize("hw", [(300, 200), (224, 386), (224, 312), (640, 512), (224, 224), (97, 333)])
def test_processors_match_transformers(hw):
    from transformers import CLIPImageProcessor, ViTImageProcessor
    img = np.random.RandomState(hw[0] + hw[1]).randint(0, 256, hw + (3,), dtype=np.uint8)
    pil = Image.fromarray(img)
    ref_c = CLIPImageProcessor()(images=[pil], return_tensors="np").pixel_values[0]
    ref_v = ViTImageProcessor()(images=[pil], return_tensors="np").pixel_values[0]
    got_c, got_v = P.clip_preprocess(img), P.vit_preprocess(img)
    assert got_c.shape == ref_c.shape == (3, 224, 224) and got_c.dtype == np.float32
    assert np.abs(got_c - ref_c).max() <= 1e-6 and np.abs(got_v - ref_v).max() <= 1e-6


@pytest.mark.parametrize("filt", ["bicubic", "bilinear"])
def test_product_tables_equal_the_oracle_tables(filt):
    """diffsensei_amd/preprocess.py builds the coefficient tables the device kernels consume; they must be the tables the
    Pillow-pinned oracle derives (the kernels then only do the integer multiply-accumulate)."""
    from diffsensei_amd.preprocess import resample_tables, shortest_edge_size
    for n_in, n_out in [(200, 224), (300, 336), (224, 224), (640, 224), (911, 551), (5, 224), (1000, 224), (333, 517)]:
        first, count, taps = resample_tables(n_in, n_out, filt)
        o_first, o_count, o_taps = P.precompute_coeffs(n_in, n_out, filt)
        assert np.array_equal(first, o_first) and np.array_equal(count, o_count) and np.array_equal(taps, o_taps)
        assert taps.dtype == np.int32 and int(np.abs(taps.astype(np.int64)).sum(1).max()) * 255 < 2 ** 31   # int32 accumulate is safe
    assert shortest_edge_size(97, 333, 224) == P.shortest_edge_size(97, 333, 224)


@pytest.mark.parametrize("mode", ["RGBA", "P", "L", "LA", "CMYK", "RGB"])
@pytest.mark.parametrize("hw", [(225, 223), (97, 333), (224, 224), (641, 479)])
def test_non_rgb_modes_and_odd_sizes_match_the_hf_processors(mode, hw):
    """the pipeline's default is the device pre-processing (`pipeline.device_preprocess`), whose host part is
    `image.convert("RGB")` -> bytes (diffsensei_amd/preprocess.py `_one`); everything after is the byte-exact resize pinned
    above.  The reference hands the PIL images to CLIPImageProcessor() / ViTImageProcessor() (reference
    src/pipelines/pipeline_diffsensei.py:125-126).  For every PIL mode a reference image can arrive in, and odd sizes, the
    chain convert("RGB") -> oracle resize/crop/normalise must equal the HF processors on the ORIGINAL image wherever they
    accept it (CLIP: do_convert_rgb; ViT: accepts 3-channel input only - for other modes the reference itself fails or
    mis-normalises, and the comparison is made on the RGB-converted image)."""
    from transformers import CLIPImageProcessor, ViTImageProcessor
    rng = np.random.RandomState(hw[0] * 7 + hw[1] + len(mode))
    base = Image.fromarray(rng.randint(0, 256, hw + (3,), dtype=np.uint8))
    if mode == "RGBA":
        im = base.copy()
        im.putalpha(Image.fromarray(rng.randint(0, 256, hw, dtype=np.uint8)))
    elif mode == "LA":
        im = base.convert("L")
        im.putalpha(Image.fromarray(rng.randint(0, 256, hw, dtype=np.uint8)))
    else:
        im = base.convert(mode)
    assert im.mode == mode
    rgb = np.array(im.convert("RGB"), dtype=np.uint8)
    got_c, got_v = P.clip_preprocess(rgb), P.vit_preprocess(rgb)
    ref_c = CLIPImageProcessor()(images=[im], return_tensors="np").pixel_values[0]          # the ORIGINAL image
    assert ref_c.shape == (3, 224, 224) and np.abs(got_c - ref_c).max() <= 1e-6, mode
    ref_v = ViTImageProcessor()(images=[im.convert("RGB")], return_tensors="np").pixel_values[0]
    assert np.abs(got_v - ref_v).max() <= 1e-6, mode
    if mode == "RGB":
        assert np.array_equal(ref_v, ViTImageProcessor()(images=[im], return_tensors="np").pixel_values[0])


def test_device_preprocess_env_switch(monkeypatch):
    """DIFFSENSEI_DEVICE_PREPROCESS=0 restores the reference's host processors without touching code."""
    from diffsensei_amd.pipeline import DiffSenseiPipeline
    from diffsensei_amd.schedulers import EulerDiscreteScheduler
    unet = type("U", (), {"config": type("C", (), {"sample_size": 128})(), "device": "cpu"})()
    monkeypatch.setenv("DIFFSENSEI_DEVICE_PREPROCESS", "0")
    assert DiffSenseiPipeline(None, None, None, None, None, EulerDiscreteScheduler(), unet, None).device_preprocess is False
    monkeypatch.delenv("DIFFSENSEI_DEVICE_PREPROCESS")
    assert DiffSenseiPipeline(None, None, None, None, None, EulerDiscreteScheduler(), unet, None).device_preprocess is True
