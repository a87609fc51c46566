"""GPU: character encoders and the whole `DiffSenseiPipeline.__call__` (tiny widths, true token counts) vs the CPU
oracle: transformers CLIP-vision / ViT-MAE in fp32, oracle Resampler, oracle sampling loop.

Tolerance: fp16 engine vs fp32 reference modules, relative L2 <= 2e-2 for encoder outputs, <= 5e-2 for the latents
after 3 full denoise steps (error compounds through CFG at guidance 7.5).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
hq = lambda t: t.half().float()


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-6)).item()


@pytest.fixture(scope="module")
def encoders(hip_lib):
    from transformers import CLIPVisionConfig, CLIPVisionModel, ViTMAEConfig, ViTMAEModel
    torch.manual_seed(0)
    clip = CLIPVisionModel(CLIPVisionConfig(hidden_size=160, intermediate_size=320, num_hidden_layers=4,
                                            num_attention_heads=2, image_size=224, patch_size=14, hidden_act="gelu")).eval()
    mae = ViTMAEModel(ViTMAEConfig(hidden_size=128, num_hidden_layers=3, num_attention_heads=2, intermediate_size=256,
                                   image_size=224, patch_size=16, mask_ratio=0.0)).eval()
    return clip, mae


def test_clip_and_mae_engines_vs_transformers(encoders):
    from diffsensei_amd.encoders import ClipVisionEngine, ViTMAEEngine
    clip, mae = encoders
    g = torch.Generator().manual_seed(1)
    px = torch.randn(3, 3, 224, 224, generator=g)
    with torch.no_grad():
        ref_c = clip(px, output_hidden_states=True).hidden_states[-2]
        ref_m = mae(px).last_hidden_state[:, 0]
    ce, me = ClipVisionEngine.from_transformers(clip, DEV), ViTMAEEngine.from_transformers(mae, DEV)
    got_c, got_m = ce.penultimate_hidden(px), me.cls_embedding(px)
    assert got_c.shape == ref_c.shape == (3, 257, 160) and got_m.shape == ref_m.shape == (3, 128)
    assert _rel(got_c, ref_c) <= 2e-2, _rel(got_c, ref_c)
    assert _rel(got_m, ref_m) <= 2e-2, _rel(got_m, ref_m)


def test_pipeline_call_vs_oracle(encoders):
    from PIL import Image
    from diffsensei_amd.pipeline import DiffSenseiPipeline
    from diffsensei_amd.resampler import Resampler
    from diffsensei_amd.schedulers import EulerDiscreteScheduler
    from diffsensei_amd.unet import UNetMangaModel
    from diffsensei_amd.unet_config import random_state_dict, tiny_config
    from oracle.pipeline_ref import sample_loop
    from oracle.resampler_ref import resampler_forward
    from oracle.scheduler_ref import EulerDiscreteOracle
    from oracle.unet_ref import UNetOracle
    clip, mae = encoders
    cfg = tiny_config()
    sd = {k: v.half() for k, v in random_state_dict(cfg, 2).items()}
    unet = UNetMangaModel(cfg, device=DEV)
    unet.load_state_dict(sd)
    rs = Resampler(dim=128, depth=2, dim_head=64, heads=2, num_queries=16, num_dummy_tokens=16, embedding_dim=160,
                   magi_embedding_dim=128, output_dim=cfg.cross_attention_dim, ff_mult=4, device=DEV).init_random(3)
    pipe = DiffSenseiPipeline(None, None, None, None, None, EulerDiscreteScheduler(), unet, clip)
    pipe.register_manga_modules(magi_image_encoder=mae, image_proj_model=rs)
    rng = np.random.RandomState(0)
    imgs = [Image.fromarray(rng.randint(0, 256, (224, 224, 3), dtype=np.uint8)) for _ in range(2)]
    ip_bbox = [[0.05, 0.10, 0.50, 0.95], [0.50, 0.10, 0.95, 0.95]]
    dialog = [[0.05, 0.02, 0.30, 0.15], [0.65, 0.02, 0.95, 0.15]]
    g = torch.Generator().manual_seed(5)
    pe = torch.randn(1, 77, cfg.cross_attention_dim, generator=g).half()
    pooled = torch.randn(1, cfg.projection_class_embeddings_input_dim - 6 * cfg.addition_time_embed_dim, generator=g).half()
    steps, ns, size = 3, 2, 128
    lat0 = torch.randn(ns, 4, size // 8, size // 8, generator=g).half()
    with pytest.raises(ValueError):
        pipe(prompt="p", ip_images=imgs, ip_bbox=ip_bbox[:1], prompt_embeds=pe, pooled_prompt_embeds=pooled,
             output_type="latent")
    results = []
    for use_graph in (False, True):
        pipe.use_graph = use_graph
        out = pipe(prompt="a manga panel", height=size, width=size, num_inference_steps=steps, guidance_scale=7.5,
                   num_samples=ns, ip_images=list(imgs), ip_bbox=[list(b) for b in ip_bbox], ip_scale=0.6,
                   dialog_bbox=[list(b) for b in dialog], latents=lat0.clone(), prompt_embeds=pe,
                   pooled_prompt_embeds=pooled, output_type="latent").images
        assert pipe.last_run_info["graph"] == use_graph
        results.append(out.clone())
    assert torch.equal(results[0], results[1])
    # ---- oracle pipeline (reference :104-154 and :294-337 restated with fp32 modules)
    clip_px = pipe._processors()[0](images=imgs + [Image.new("RGB", (224, 224))] * 2, return_tensors="pt").pixel_values
    magi_px = pipe._processors()[1](images=imgs + [Image.new("RGB", (224, 224))] * 2, return_tensors="pt").pixel_values
    with torch.no_grad():
        ce = clip(clip_px, output_hidden_states=True).hidden_states[-2].unsqueeze(0)
        me = mae(magi_px).last_hidden_state[:, 0].unsqueeze(0)
        ce[0, 2:], me[0, 2:] = 0, 0
        rsd = {k: v.float().cpu() for k, v in rs.state_dict().items()}
        img = resampler_forward(rsd, ce, me, 2, 64)
        neg = resampler_forward(rsd, torch.zeros_like(ce), torch.zeros_like(me), 2, 64)
        enc = torch.cat([torch.cat([torch.zeros_like(pe.float()).repeat(ns, 1, 1), pe.float().repeat(ns, 1, 1)]),
                         torch.cat([neg.repeat(ns, 1, 1), img.repeat(ns, 1, 1)])], dim=1)
        te = torch.cat([torch.zeros(ns, pooled.shape[1]), pooled.float().repeat(ns, 1)])
        tid = torch.tensor([[size, size, 0, 0, size, size]] * (2 * ns), dtype=torch.float32)
        bbox = torch.zeros(2 * ns, 4, 4)
        bbox[ns:, 0], bbox[ns:, 1] = torch.tensor(ip_bbox[0]), torch.tensor(ip_bbox[1])
        db = torch.zeros(2 * ns, 8, 4, dtype=torch.float16)
        db[ns:, 0], db[ns:, 1] = torch.tensor(dialog[0]).half(), torch.tensor(dialog[1]).half()
        sch = EulerDiscreteOracle().set_timesteps(steps)
        ref = sample_loop(UNetOracle(cfg, sd, q=hq), EulerDiscreteOracle(), hq(lat0.float() * sch.init_noise_sigma),
                          hq(enc), hq(te), tid, bbox, db, 7.5, steps, 0.6, q=hq)
    assert _rel(results[0], ref) <= 5e-2, _rel(results[0], ref)
