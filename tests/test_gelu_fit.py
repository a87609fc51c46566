"""ds_gelu_erf (csrc/ds_common.h) on the host: the constants are parsed out of the header and the device's f32 evaluation
order is replayed in numpy on ALL finite f16 inputs against the exact erf-GELU (the activation of diffusers' GEGLU [3P],
reached from /root/reference/src/models/unet.py:244-338, and of the CLIP-H / bigG MLPs).  Not a GPU test: it pins the
polynomial itself; tests/test_gpu_ops.py and tests/test_gpu_ln_fusion.py check the kernels that use it."""
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
scipy_special = pytest.importorskip("scipy.special")
import fit_gelu  # noqa: E402


def _header_coefficients():
    src = open(os.path.join(ROOT, "diffsensei_amd", "csrc", "ds_common.h")).read()
    body = src[src.index("float ds_gelu_erf(float x)"):src.index("f32x2 ds_gelu_erf2")]
    first = re.search(r"float q = (-?[0-9.e+-]+)f;", body).group(1)
    rest = re.findall(r"q = fmaf\(q, a, (-?[0-9.e+-]+)f\);", body)
    clamp = float(re.search(r"fminf\(fabsf\(x\), ([0-9.]+)f\)", body).group(1))
    pair = src[src.index("f32x2 ds_gelu_erf2"):src.index("// Source index of nearest-neighbour")]
    pair_c = [re.search(r"f32x2 q = k\((-?[0-9.e+-]+)f\);", pair).group(1)] + re.findall(r"a, k\((-?[0-9.e+-]+)f\)\);", pair)
    return [float(first)] + [float(v) for v in rest], clamp, [float(v) for v in pair_c]


def test_scalar_and_paired_forms_hold_the_same_constants():
    horner, clamp, pair = _header_coefficients()
    assert len(horner) == 11 and horner == pair and clamp == fit_gelu.CLAMP


def test_header_constants_are_the_fit():
    horner, _, _ = _header_coefficients()
    co, dq = fit_gelu.fit()
    assert dq < 1e-6
    np.testing.assert_allclose(horner[::-1], co, rtol=2e-8, atol=0)


def test_f16_rounding_of_the_device_form_on_every_f16_input():
    horner, _, _ = _header_coefficients()
    co = np.array(horner[::-1])
    for name, x in fit_gelu.samples().items():
        rate, ulp = fit_gelu.misround(fit_gelu.gelu_dev(x, co), x)
        assert ulp <= 1 and rate < 5e-4, (name, rate, ulp)
    allh = fit_gelu.samples()["all finite f16"]
    g = fit_gelu.gelu_dev(allh, co)
    assert np.isfinite(g).all()
    x = allh.astype(np.float32)
    assert (g[x > 8] == x[x > 8]).all() and (np.abs(g[x < -8]) < 3e-8).all()   # clamp region: identity / below an f16 subnormal
    assert (g[x == 0] == 0).all()
