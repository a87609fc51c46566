"""TEST INFRASTRUCTURE ONLY — CPU restatement of the character-reference pre-processing (SURVEY.md §8(f) rank 4,
"image pre-processing on GPU ... currently PIL on CPU", reference src/pipelines/pipeline_diffsensei.py:125-126):

    clip_image_processor = CLIPImageProcessor()   # RGB -> shortest edge 224 (PIL BICUBIC) -> centre crop 224 ->
    magi_image_processor = ViTImageProcessor()    #   /255 -> (x - mean) / std, CHW fp32
                                                  # ViT: 224 x 224 (PIL BILINEAR) -> /255 -> (x - 0.5) / 0.5

The arithmetic lives in third-party code that IS installed here: Pillow 12.2.0 `Image.resize` (libImaging
Resample.c: separable, antialiased, 8-bit fixed point, 22 coefficient bits, horizontal pass then vertical pass with an
8-bit intermediate image) and transformers 5.15 image processors.  This file restates that integer algorithm in numpy;
tests/test_oracle_preprocess.py pins it BIT-EXACT against Pillow on random images and against the two transformers
processors end to end.  It is the oracle a device implementation of the pre-processing will be held to (not built yet).
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _bicubic(x: np.ndarray) -> np.ndarray:
    a = -0.5
    x = np.abs(x)
    return np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1, np.where(x < 2.0, (((x - 5) * x + 8) * x - 4) * a, 0.0))


def _bilinear(x: np.ndarray) -> np.ndarray:
    x = np.abs(x)
    return np.where(x < 1.0, 1.0 - x, 0.0)


FILTERS = {"bicubic": (_bicubic, 2.0), "bilinear": (_bilinear, 1.0)}


def precompute_coeffs(in_size: int, out_size: int, filt: str) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the whole-image box: per output index the first source
    index, the tap count and the int32 taps (scaled by 2^22, rounded half away from zero)."""
    fn, fsupport = FILTERS[filt]
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = fsupport * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    xmin = np.zeros(out_size, np.int64)
    cnt = np.zeros(out_size, np.int64)
    kk = np.zeros((out_size, ksize), np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        lo = max(int(center - support + 0.5), 0)
        hi = min(int(center + support + 0.5), in_size)
        n = hi - lo
        w = fn((np.arange(n) + lo - center + 0.5) * ss)
        ww = 0.0
        for v in w:                      # same left-to-right double accumulation as the C loop
            ww += float(v)
        if ww != 0.0:
            w = w / ww
        pre = w * (1 << PRECISION_BITS)
        kk[xx, :n] = np.where(pre < 0, (-0.5 + pre).astype(np.int64), (0.5 + pre).astype(np.int64))  # C cast truncates
        xmin[xx], cnt[xx] = lo, n
    return xmin, cnt, kk


def _resample_axis0(img: np.ndarray, out_size: int, filt: str) -> np.ndarray:
    """One 8-bit pass along axis 0 of a [N, ...] uint8 array."""
    xmin, cnt, kk = precompute_coeffs(img.shape[0], out_size, filt)
    out = np.empty((out_size,) + img.shape[1:], np.uint8)
    src = img.astype(np.int64)
    for xx in range(out_size):
        n = cnt[xx]
        taps = kk[xx, :n].reshape((n,) + (1,) * (img.ndim - 1))
        acc = (1 << (PRECISION_BITS - 1)) + (src[xmin[xx]:xmin[xx] + n] * taps).sum(0)
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def pil_resize_u8(img: np.ndarray, out_h: int, out_w: int, filt: str) -> np.ndarray:
    """`Image.fromarray(img).resize((out_w, out_h), BICUBIC|BILINEAR)` for uint8 [H,W,C]: horizontal pass (skipped when
    the width is unchanged), then vertical pass (skipped when the height is unchanged)."""
    h, w = img.shape[:2]
    if out_w != w:
        img = np.swapaxes(_resample_axis0(np.swapaxes(img, 0, 1), out_w, filt), 0, 1)
    if out_h != h:
        img = _resample_axis0(img, out_h, filt)
    return np.ascontiguousarray(img)


def shortest_edge_size(h: int, w: int, size: int = 224) -> Tuple[int, int]:
    """transformers get_resize_output_image_size(default_to_square=False): short side -> size, long side truncated."""
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def clip_preprocess(img: np.ndarray) -> np.ndarray:
    """CLIPImageProcessor() on one RGB uint8 [H,W,3] image -> fp32 [3,224,224]."""
    h, w = img.shape[:2]
    nh, nw = shortest_edge_size(h, w, 224)
    r = pil_resize_u8(img, nh, nw, "bicubic")
    top, left = (nh - 224) // 2, (nw - 224) // 2
    r = r[top:top + 224, left:left + 224].astype(np.float32)
    x = r * np.float32(1 / 255)
    x = (x - np.asarray(CLIP_MEAN, np.float32)) / np.asarray(CLIP_STD, np.float32)
    return np.ascontiguousarray(x.transpose(2, 0, 1))


def vit_preprocess(img: np.ndarray) -> np.ndarray:
    """ViTImageProcessor() on one RGB uint8 [H,W,3] image -> fp32 [3,224,224]."""
    r = pil_resize_u8(img, 224, 224, "bilinear").astype(np.float32)
    x = (r * np.float32(1 / 255) - np.float32(0.5)) / np.float32(0.5)
    return np.ascontiguousarray(x.transpose(2, 0, 1))
