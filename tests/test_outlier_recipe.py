"""CPU: the outlier-regime weight recipe (tests/_outliers.py) does what its docstring says on the oracle side - the transformer
streams carry |h| of a few hundred in a few percent of their channels, the fp16-storage oracle stays finite and close to the
fp32 one - so the GPU test built on it (tests/test_gpu_outlier_magnitudes.py) compares against a meaningful reference."""
import torch


def test_outlier_recipe_regime_on_the_tiny_config():
    from diffsensei_amd.unet_config import random_state_dict, tiny_config
    from oracle.unet_ref import UNetOracle
    from tests._outliers import RegimeProbe, make_outlier_state_dict
    from tests.test_gpu_unet import _inputs
    cfg = tiny_config()
    sd = {k: v.half() for k, v in random_state_dict(cfg, 0).items()}
    so = make_outlier_state_dict(sd, frac=0.04)
    assert all(torch.equal(v, v.half().float()) for v in so.values()), "weights must be fp16-representable"
    x, enc, te, tid, bbox, db = _inputs(cfg, 2, 16, 16)
    hq = lambda t: t.half().float()
    o16, o32 = UNetOracle(cfg, so, q=hq), UNetOracle(cfg, so)
    probe = RegimeProbe(o16)
    with torch.no_grad():
        r16 = o16.forward(x, 801.0, enc, te, tid, bbox, 1.0, db)
        probe.close()
        r32 = o32.forward(x, 801.0, enc, te, tid, bbox, 1.0, db)
    assert torch.isfinite(r16).all() and torch.isfinite(r32).all()
    assert 100.0 <= probe.max_h <= 4000.0 and probe.frac_big >= 0.02, (probe.max_h, probe.frac_big)
    assert probe.max_logit >= 30.0
    assert float((r16 - r32).norm() / r32.norm()) <= 5e-3
    # the plain weights stay O(1) under the same probe
    o = UNetOracle(cfg, {k: v.float() for k, v in sd.items()})
    p2 = RegimeProbe(o)
    with torch.no_grad():
        o.forward(x, 801.0, enc, te, tid, bbox, 1.0, db)
    p2.close()
    assert p2.max_h < 20.0 and p2.frac_big == 0.0
