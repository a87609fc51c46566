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


# ------------------------------------------------------------------------- independence of the oracle's topology
def _product_program(cfg):
    """The product's `build_topology` flattened into the step vocabulary of oracle/unet_topology_ref.py."""
    from diffsensei_amd.unet_config import build_topology
    topo, prog = build_topology(cfg), [("push",)]
    for blk in topo.down:
        for j, r in enumerate(blk["resnets"]):
            prog.append(("resnet", r.prefix, r.cin, r.cout))
            assert r.has_shortcut == (r.cin != r.cout)
            if blk["attns"]:
                a = blk["attns"][j]
                prog.append(("attn", a.prefix, a.channels, a.depth, a.heads))
            prog.append(("push",))
        if blk["downsample"]:
            prog += [("downsample", blk["downsample"], blk["resnets"][-1].cout), ("push",)]
    m = topo.mid
    prog += [("resnet", m["resnets"][0].prefix, m["resnets"][0].cin, m["resnets"][0].cout),
             ("attn", m["attns"][0].prefix, m["attns"][0].channels, m["attns"][0].depth, m["attns"][0].heads),
             ("resnet", m["resnets"][1].prefix, m["resnets"][1].cin, m["resnets"][1].cout)]
    for blk in topo.up:
        for j, r in enumerate(blk["resnets"]):
            prog += [("pop_cat",), ("resnet", r.prefix, r.cin, r.cout)]
            assert r.has_shortcut == (r.cin != r.cout)
            if blk["attns"]:
                a = blk["attns"][j]
                prog.append(("attn", a.prefix, a.channels, a.depth, a.heads))
        if blk["upsample"]:
            prog.append(("upsample", blk["upsample"], blk["resnets"][-1].cout))
    return prog


def test_oracle_topology_is_independent_and_agrees_with_the_product():
    """oracle/unet_ref.py no longer imports the product's topology table: the oracle derives the
    block order / widths / skip pairing by simulating the channel flow; the product computes them from formulas.  They
    must describe the same network, for the SDXL config, the tiny config and an asymmetric one."""
    import inspect
    import oracle.unet_ref as U
    import oracle.unet_topology_ref as T
    from diffsensei_amd.unet_config import UNetMangaConfig, param_shapes, sdxl_config
    assert "build_topology" not in inspect.getsource(U).replace("unet_config.build_topology`", "")
    assert "import" not in "".join(l for l in inspect.getsource(T).splitlines() if "diffsensei_amd" in l and "import" in l)
    odd = UNetMangaConfig(block_out_channels=(64, 192, 128, 256), down_block_types=("DownBlock2D", "CrossAttnDownBlock2D",
                          "DownBlock2D", "CrossAttnDownBlock2D"), up_block_types=("CrossAttnUpBlock2D", "UpBlock2D",
                          "CrossAttnUpBlock2D", "UpBlock2D"), transformer_layers_per_block=(1, 3, 1, 2),
                          attention_head_dim=(1, 3, 2, 4), layers_per_block=1)
    for cfg in (sdxl_config(), tiny_config(), odd):
        assert T.unet_program(T.config_dict(cfg)) == _product_program(cfg)
        assert dict(param_shapes(cfg)) == T.program_param_shapes(T.config_dict(cfg), manga=True)
    assert T.config_dict(sdxl_config()) == T.SDXL_BASE_UNET_CONFIG


def test_sdxl_unet_public_anchors():
    """Facts about the public SDXL-base UNet checkpoint (stabilityai/stable-diffusion-xl-base-1.0, unet/): 1680 tensors,
    2 567 463 684 parameters, and the up-path conv1 input widths that encode the skip pairing."""
    import math
    import oracle.unet_topology_ref as T
    base = T.program_param_shapes(T.SDXL_BASE_UNET_CONFIG)
    assert len(base) == 1680
    assert sum(math.prod(s) for s in base.values()) == 2_567_463_684
    known = {"up_blocks.0.resnets.0.conv1.weight": (1280, 2560, 3, 3), "up_blocks.0.resnets.2.conv1.weight": (1280, 1920, 3, 3),
             "up_blocks.1.resnets.0.conv1.weight": (640, 1920, 3, 3), "up_blocks.1.resnets.1.conv1.weight": (640, 1280, 3, 3),
             "up_blocks.1.resnets.2.conv1.weight": (640, 960, 3, 3), "up_blocks.2.resnets.0.conv1.weight": (320, 960, 3, 3),
             "up_blocks.2.resnets.1.conv1.weight": (320, 640, 3, 3), "up_blocks.2.resnets.2.conv1.weight": (320, 640, 3, 3),
             "down_blocks.1.resnets.0.conv_shortcut.weight": (640, 320, 1, 1),
             "down_blocks.2.attentions.1.transformer_blocks.9.attn2.to_k.weight": (1280, 2048),
             "down_blocks.1.attentions.0.transformer_blocks.1.ff.net.0.proj.weight": (5120, 640),
             "mid_block.attentions.0.transformer_blocks.9.ff.net.2.weight": (1280, 5120),
             "add_embedding.linear_1.weight": (1280, 2816), "time_embedding.linear_1.weight": (1280, 320)}
    for k, shp in known.items():
        assert base[k] == shp, (k, base[k])
    assert "down_blocks.0.attentions.0.norm.weight" not in base          # first level is attention-free
    assert "down_blocks.2.downsamplers.0.conv.weight" not in base and "up_blocks.2.upsamplers.0.conv.weight" not in base
    manga = T.program_param_shapes(T.SDXL_BASE_UNET_CONFIG, manga=True)
    assert len(manga) - len(base) == 2 * 70 + 1                           # 70 IP K/V pairs + dialog_bbox_embedding


# ------------------------------------------------------------------------- known-answer constants of the [3P] schedulers
def _sd_schedule_f64():
    """The Stable-Diffusion 'scaled_linear' schedule in float64 from its published definition (independent of both the
    product's and the oracle's scheduler classes)."""
    import numpy as np
    betas = np.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=np.float64) ** 2
    ac = np.cumprod(1.0 - betas)
    return ac, np.sqrt((1 - ac) / ac)


def test_scheduler_known_answers():
    """Anchors outside this repository: the SD/SDXL discrete sigmas span 0.0292 .. 14.6146 (k-diffusion's sigma_min /
    sigma_max for this beta schedule); diffusers' "leading" spacing with steps_offset 1 gives 981..1 (50 steps),
    951..1 (20), 958..1 step 33 (30); EulerDiscrete init_noise_sigma = sqrt(max sigma^2 + 1) on the chosen grid;
    DDIM's final alpha (set_alpha_to_one False) = alphas_cumprod[0] = 1 - 0.00085; SDXL VAE scaling 0.13025."""
    import numpy as np
    from diffsensei_amd.schedulers import DDIMScheduler, EulerDiscreteScheduler
    from diffsensei_amd.vae import VaeConfig
    ac, sig = _sd_schedule_f64()
    assert abs(sig[0] - 0.0292) < 5e-5 and abs(sig[-1] - 14.6146) < 5e-4
    assert abs(ac[0] - 0.99915) < 1e-9 and abs(ac[-1] - 0.0046598) < 5e-7
    for n, first, step in ((50, 981, 20), (20, 951, 50), (30, 958, 33)):
        expect = np.arange(first, 0, -step, dtype=np.float64)
        assert len(expect) == n and expect[-1] == 1
        for S in (EulerDiscreteScheduler(), EulerDiscreteOracle()):
            S.set_timesteps(n)
            ts = np.asarray(S.timesteps, dtype=np.float64)
            assert ts.tolist() == expect.tolist()
            want = np.interp(expect, np.arange(1000), sig)           # integer timesteps: interpolation is exact
            np.testing.assert_allclose(np.asarray(S.sigmas[:-1], dtype=np.float64), want, rtol=1e-5)  # fp32 cumprod like diffusers
            assert S.sigmas[-1] == 0.0
            assert abs(S.init_noise_sigma - (want[0] ** 2 + 1) ** 0.5) < 2e-5
        D, DO = DDIMScheduler(), DDIMOracle()
        D.set_timesteps(n)
        DO.set_timesteps(n)
        assert np.asarray(D.timesteps).tolist() == expect.tolist() == np.asarray(DO.timesteps).tolist()
        tab = D.coef_table(7.5)
        np.testing.assert_allclose(tab[:, 2], np.sqrt(ac[expect.astype(int)]), rtol=1e-5)
        np.testing.assert_allclose(tab[:, 3], np.sqrt(1 - ac[expect.astype(int)]), rtol=1e-5)
        prev = expect.astype(int) - 1000 // n
        a_prev = np.where(prev >= 0, ac[np.clip(prev, 0, None)], ac[0])
        np.testing.assert_allclose(tab[:, 4], np.sqrt(a_prev), rtol=1e-5)
        assert abs(tab[-1, 4] ** 2 - 0.99915) < 1e-6                  # last step lands on final_alpha_cumprod
        assert D.init_noise_sigma == 1.0 and DO.init_noise_sigma == 1.0
    e50 = EulerDiscreteScheduler()
    e50.set_timesteps(50)
    assert abs(e50.sigmas[0] - sig[981]) < 1e-4 and 12.5 < e50.init_noise_sigma < 13.5
    assert VaeConfig().scaling_factor == 0.13025


def test_oracle_handles_latent_sizes_that_are_not_multiples_of_four():
    """diffusers `forward_upsample_size`: the up path resizes to the skip tensors' sizes; the oracle follows it."""
    cfg = tiny_config()
    sd = random_state_dict(cfg, 0)
    u = UNetOracle(cfg, sd)
    x, enc, te, tid, bbox, db = _inputs(cfg, 1, 18, 13)
    with torch.no_grad():
        y = u.forward(x, 500.0, enc, te, tid, bbox, 18 / 13, db)
        assert y.shape == x.shape and torch.isfinite(y).all()
        # multiples of 4: resizing to the skip's size and plain x2 are the same thing
        x2, enc2, te2, tid2, bbox2, db2 = _inputs(cfg, 1, 16, 12)
        a = u.forward(x2, 500.0, enc2, te2, tid2, bbox2, 16 / 12, db2)
    assert a.shape == x2.shape
