"""CPU: self-consistency of the UNet / scheduler / loop oracle on the tiny config."""
import torch

from diffsensei_amd.unet_config import random_state_dict, tiny_config
from oracle.pipeline_ref import sample_loop
from oracle.scheduler_ref import DDIMOracle, EulerDiscreteOracle, cfg_combine
from oracle.unet_ref import UNetOracle, encode_dialog_bbox, timestep_sinusoid


def _inputs(cfg, B=2, H=16, W=16, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 4, H, W, generator=g)
    enc = torch.randn(B, cfg.num_text_tokens + cfg.num_ip_tokens, cfg.cross_attention_dim, generator=g)
    pooled = cfg.projection_class_embeddings_input_dim - 6 * cfg.addition_time_embed_dim
    te = torch.randn(B, pooled, generator=g)
    tid = torch.tensor([[H * 8, W * 8, 0, 0, H * 8, W * 8]] * B, dtype=torch.float32)
    bbox = torch.zeros(B, 4, 4)
    bbox[B // 2:, 0] = torch.tensor([0.05, 0.10, 0.50, 0.95])
    bbox[B // 2:, 1] = torch.tensor([0.50, 0.10, 0.95, 0.95])
    db = torch.zeros(B, 8, 4)
    db[B // 2:, 0] = torch.tensor([0.05, 0.02, 0.30, 0.15])
    db[B // 2:, 1] = torch.tensor([0.65, 0.02, 0.95, 0.15])
    return x, enc, te, tid, bbox, db


def test_unet_oracle_runs_and_conditions_matter():
    cfg = tiny_config()
    sd = random_state_dict(cfg, 0)
    u = UNetOracle(cfg, sd)
    x, enc, te, tid, bbox, db = _inputs(cfg)
    with torch.no_grad():
        y = u.forward(x, 801.0, enc, te, tid, bbox, 1.0, db)
        assert y.shape == x.shape and torch.isfinite(y).all()
        y2 = u.forward(x, 801.0, enc, te, tid, bbox, 1.0, db)
        assert torch.equal(y, y2)
        # every conditioning input reaches the output
        assert not torch.allclose(y, u.forward(x, 401.0, enc, te, tid, bbox, 1.0, db))
        assert not torch.allclose(y, u.forward(x, 801.0, enc, te, tid, bbox, 1.0, None))
        b2 = bbox.clone()
        b2[1, 0] = torch.tensor([0.0, 0.0, 1.0, 1.0])
        assert not torch.allclose(y[1], u.forward(x, 801.0, enc, te, tid, b2, 1.0, db)[1])
        # batch items are independent
        assert torch.allclose(y[:1], u.forward(x[:1], 801.0, enc[:1], te[:1], tid[:1], bbox[:1], 1.0, db[:1]), atol=1e-5)
        # fp16-storage emulation stays close to pure fp32
        uh = UNetOracle(cfg, {k: v.half().float() for k, v in sd.items()}, q=lambda t: t.half().float())
        yh = uh.forward(x, 801.0, enc, te, tid, bbox, 1.0, db)
        rel = (yh - y).norm() / y.norm()
        assert rel < 3e-2, rel


def test_dialog_bbox_assign_semantics():
    s = torch.zeros(1, 4, 8, 8)
    emb = torch.tensor([1.0, 2.0, 3.0, 4.0])
    db = torch.tensor([[[0.0, 0.0, 0.5, 0.5], [0.25, 0.25, 0.75, 0.75], [0.9, 0.9, 2.0, 2.0], [0, 0, 0, 0]]])
    out = encode_dialog_bbox(s, db, emb)
    # overlapping boxes ASSIGN (no accumulation): value is emb wherever any box covers
    assert out[0, :, 2, 2].tolist() == [1.0, 2.0, 3.0, 4.0]
    assert out[0, :, 5, 5].tolist() == [1.0, 2.0, 3.0, 4.0]
    assert out[0, :, 7, 7].tolist() == [1.0, 2.0, 3.0, 4.0]      # clamped box
    assert out[0, :, 6, 1].abs().sum() == 0


def test_sinusoid_flip_and_shape():
    e = timestep_sinusoid(torch.tensor([0.0, 10.0]), 320)
    assert e.shape == (2, 320)
    assert torch.allclose(e[0, :160], torch.ones(160)) and torch.allclose(e[0, 160:], torch.zeros(160))  # cos first


def test_scheduler_oracles():
    eu = EulerDiscreteOracle().set_timesteps(20)
    assert eu.timesteps[0] == 951.0 and eu.timesteps[-1] == 1.0 and eu.sigmas[-1] == 0.0
    x = torch.randn(1, 4, 8, 8)
    e = torch.randn(1, 4, 8, 8)
    y = eu.step(e, 3, x)
    assert torch.allclose(y, x + e * (eu.sigmas[4] - eu.sigmas[3]), atol=1e-4)
    dd = DDIMOracle().set_timesteps(50)
    assert dd.timesteps[0] == 981 and dd.timesteps[-1] == 1
    # a perfect epsilon prediction recovers x0 along the DDIM path
    x0 = torch.randn(1, 4, 8, 8)
    a_t = dd.alphas_cumprod[dd.timesteps[0]]
    xt = a_t.sqrt() * x0 + (1 - a_t).sqrt() * e
    xt = dd.step(e, 0, xt)
    a_p = dd.alphas_cumprod[dd.timesteps[1]]
    assert torch.allclose(xt, a_p.sqrt() * x0 + (1 - a_p).sqrt() * e, atol=1e-5)
    nz = torch.cat([e, 2 * e])
    assert torch.allclose(cfg_combine(nz, 7.5), e + 7.5 * e)


def test_sample_loop_oracle_tiny():
    cfg = tiny_config()
    sd = random_state_dict(cfg, 1)
    u = UNetOracle(cfg, sd)
    x, enc, te, tid, bbox, db = _inputs(cfg, B=2, H=8, W=8)
    sch = EulerDiscreteOracle().set_timesteps(4)
    lat = x[:1] * sch.init_noise_sigma
    with torch.no_grad():
        out = sample_loop(u, EulerDiscreteOracle(), lat, enc, te, tid, bbox, db, 7.5, 4, 0.6)
    assert out.shape == lat.shape and torch.isfinite(out).all()
