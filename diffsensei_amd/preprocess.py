"""Character-reference pre-processing on the device.

Replaces `CLIPImageProcessor()(images=...)` / `ViTImageProcessor()(images=...)` of the reference
(src/pipelines/pipeline_diffsensei.py:125-126; scripts/demo/gradio.py:91-92), which run Pillow's resize on the host: the
RGB bytes go up once, `csrc/preprocess.hip` does the two 8-bit resample passes, the centre crop, the 1/255 rescale and the
normalisation, and the fp32 CHW pixel tensors the encoders consume stay on the device.

The host computes only the coefficient tables — Pillow's `precompute_coeffs` + `normalize_coeffs_8bpc` (libImaging
Resample.c) in the same double arithmetic, cached per (input size, output size, filter) — so the device result equals
Pillow's byte for byte (`tests/test_gpu_preprocess.py`; the oracle, pinned against Pillow itself, is
oracle/image_preprocess_ref.py).  No CPU fallback: without the HIP library the constructor raises.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import check

Tensor = torch.Tensor
PRECISION_BITS = 32 - 8 - 2
CLIP_MEAN, CLIP_STD = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)
VIT_MEAN, VIT_STD = (0.5, 0.5, 0.5), (0.5, 0.5, 0.5)


def _bicubic(x: float) -> float:
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _bilinear(x: float) -> float:
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


_FILTERS = {"bicubic": (_bicubic, 2.0), "bilinear": (_bilinear, 1.0)}


def resample_tables(in_size: int, out_size: int, filt: str) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(first[out], count[out], taps[out, ksize]) int32 — Resample.c precompute_coeffs for the whole-image box followed by
    normalize_coeffs_8bpc (taps scaled by 2^22 and rounded half away from zero by a truncating cast)."""
    fn, fsupport = _FILTERS[filt]
    scale = in_size / out_size
    filterscale = scale if scale > 1.0 else 1.0
    support = fsupport * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    first = np.zeros(out_size, np.int32)
    count = np.zeros(out_size, np.int32)
    taps = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    one = float(1 << PRECISION_BITS)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        lo = int(center - support + 0.5)
        lo = lo if lo > 0 else 0
        hi = int(center + support + 0.5)
        hi = hi if hi < in_size else in_size
        w = [fn((x + lo - center + 0.5) * ss) for x in range(hi - lo)]
        ww = 0.0
        for v in w:
            ww += v
        for x, v in enumerate(w):
            if ww != 0.0:
                v = v / ww
            taps[xx, x] = int(-0.5 + v * one) if v < 0 else int(0.5 + v * one)
        first[xx], count[xx] = lo, hi - lo
    return first, count, taps


def shortest_edge_size(h: int, w: int, size: int) -> Tuple[int, int]:
    """transformers `get_resize_output_image_size(..., default_to_square=False)`: short side -> size, long side truncated."""
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


class DevicePreprocessor:
    """`clip(images)` / `vit(images)`: lists of PIL images (any mode/size) -> fp32 [n,3,224,224] device tensors."""

    def __init__(self, device, size: int = 224):
        self.lib = _lib.load()
        self.dev = torch.device(device)
        self.size = int(size)
        self._tables: Dict[tuple, tuple] = {}
        self.keep_bytes = False          # tests: also return the cropped uint8 image of the last call
        self.last_bytes: List[Tensor] = []

    def _table(self, in_size: int, out_size: int, filt: str):
        key = (in_size, out_size, filt)
        t = self._tables.get(key)
        if t is None:
            first, count, taps = resample_tables(in_size, out_size, filt)
            t = tuple(torch.from_numpy(a).to(self.dev) for a in (first, count, taps)) + (taps.shape[1],)
            self._tables[key] = t
        return t

    def _one(self, image, out_hw: Tuple[int, int], filt: str, crop: bool, mean: Sequence[float], std: Sequence[float]) -> Tensor:
        rgb = np.array(image.convert("RGB"), dtype=np.uint8)             # do_convert_rgb; bytes only (own, writable copy)
        h, w = rgb.shape[:2]
        src = torch.from_numpy(np.ascontiguousarray(rgb)).to(self.dev)
        nh, nw = out_hw
        S = self.size
        top, left = ((nh - S) // 2, (nw - S) // 2) if crop else (0, 0)
        st = torch.cuda.current_stream(self.dev).cuda_stream
        fh, ch, th, kh = self._table(w, nw, filt)
        tmp = torch.empty((h, nw, 3), dtype=torch.uint8, device=self.dev)
        check(self.lib.ds_resize_h_u8(src.data_ptr(), h, w, fh.data_ptr(), ch.data_ptr(), th.data_ptr(), kh, nw,
                                      tmp.data_ptr(), st), "ds_resize_h_u8")
        fv, cv, tv, kv = self._table(h, nh, filt)
        out = torch.empty((3, S, S), dtype=torch.float32, device=self.dev)
        by = torch.empty((S, S, 3), dtype=torch.uint8, device=self.dev) if self.keep_bytes else None
        m3, s3 = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
        check(self.lib.ds_resize_v_norm_u8(tmp.data_ptr(), h, nw, fv.data_ptr(), cv.data_ptr(), tv.data_ptr(), kv, top, left,
                                           S, S, 1.0 / 255.0, m3, s3, out.data_ptr(), None if by is None else by.data_ptr(),
                                           st), "ds_resize_v_norm_u8")
        if by is not None:
            self.last_bytes.append(by)
        return out

    def clip(self, images: Sequence) -> Tensor:
        """CLIPImageProcessor(): shortest edge -> 224 (BICUBIC), centre crop 224, /255, CLIP mean/std."""
        self.last_bytes = []
        outs = []
        for im in images:
            w, h = im.size
            outs.append(self._one(im, shortest_edge_size(h, w, self.size), "bicubic", True, CLIP_MEAN, CLIP_STD))
        return torch.stack(outs)

    def vit(self, images: Sequence) -> Tensor:
        """ViTImageProcessor(): 224 x 224 (BILINEAR), /255, mean = std = 0.5."""
        self.last_bytes = []
        return torch.stack([self._one(im, (self.size, self.size), "bilinear", False, VIT_MEAN, VIT_STD) for im in images])
