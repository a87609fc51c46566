"""GPU: character encoders and the whole `DiffSenseiPipeline.__call__` (tiny widths, true token counts) vs the CPU
oracle: transformers CLIP-vision / ViT-MAE in fp32, oracle Resampler, oracle sampling loop.

Tolerance (round 6: <= ~3-4x measured, every value logged by tests/_gates.gate): fp16 engine vs fp32 reference modules,
relative L2 <= 4e-3 for the character-encoder outputs (measured 0.7-1.1e-3), <= 1.2e-2 for the latents after 3 full denoise
steps (measured 3.7e-3; the error compounds through CFG at guidance 7.5).  Until round 5: 2e-2 / 5e-2.
"""
import numpy as np
import pytest
import torch

from tests._gates import gate

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
    gate("test_gpu_pipeline:1 " + '_rel(got_c, ref_c)', _rel(got_c, ref_c), 4e-3)
    gate("test_gpu_pipeline:2 " + '_rel(got_m, ref_m)', _rel(got_m, ref_m), 4e-3)


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
    gate("test_gpu_pipeline:3 " + '_rel(results[0], ref)', _rel(results[0], ref), 1.2e-2)


class _FakeTokenizer:
    """Deterministic stand-in for CLIPTokenizer (no vocabulary files offline): hashes words to ids, BOS/EOS/pad like CLIP."""
    model_max_length = 77

    def __init__(self, vocab=1000, bos=998, eos=999):
        self.vocab, self.bos, self.eos = vocab, bos, eos

    def __call__(self, text, padding=None, max_length=77, truncation=True, return_tensors="pt"):
        words = [1 + (sum(ord(c) * (i + 1) for i, c in enumerate(w)) % (self.bos - 2)) for w in text.split()]
        ids = [self.bos] + words[: max_length - 2] + [self.eos]
        ids += [self.eos] * (max_length - len(ids))
        return type("Enc", (), {"input_ids": torch.tensor([ids])})()


def test_text_encoder_engines_and_encode_prompt(hip_lib):
    """SDXL prompt path (reference :237-245): both CLIP text encoders on the HIP engine vs transformers fp32."""
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection
    from diffsensei_amd.encoders import ClipTextEngine
    from diffsensei_amd.pipeline import DiffSenseiPipeline
    from diffsensei_amd.schedulers import EulerDiscreteScheduler
    from diffsensei_amd.unet import UNetMangaModel
    from diffsensei_amd.unet_config import tiny_config
    torch.manual_seed(0)
    c1 = CLIPTextConfig(vocab_size=1000, hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2,
                        max_position_embeddings=77, hidden_act="quick_gelu", eos_token_id=999, bos_token_id=998, pad_token_id=0)
    c2 = CLIPTextConfig(vocab_size=1000, hidden_size=192, intermediate_size=384, num_hidden_layers=4, num_attention_heads=3,
                        max_position_embeddings=77, hidden_act="gelu", projection_dim=128, eos_token_id=999,
                        bos_token_id=998, pad_token_id=0)
    te1, te2 = CLIPTextModel(c1).eval(), CLIPTextModelWithProjection(c2).eval()
    tok = _FakeTokenizer()
    ids = torch.cat([tok("a young man holding a baby on his back").input_ids, tok("two men talking").input_ids])
    with torch.no_grad():
        r1, r2 = te1(ids, output_hidden_states=True), te2(ids, output_hidden_states=True)
    e1, e2 = ClipTextEngine.from_transformers(te1, DEV), ClipTextEngine.from_transformers(te2, DEV)
    h1, last1 = e1.encode(ids)
    h2, pooled2 = e2.encode(ids)
    assert _rel(h1, r1.hidden_states[-2]) <= 2e-2 and _rel(last1, r1[0]) <= 2e-2
    assert _rel(h2, r2.hidden_states[-2]) <= 2e-2 and _rel(pooled2, r2[0]) <= 2e-2
    assert pooled2.shape == (2, 128)
    # through the pipeline's encode_prompt, CFG with the empty negative prompt forced to zeros
    unet = UNetMangaModel(tiny_config(), device=DEV)
    pipe = DiffSenseiPipeline(None, te1, te2, tok, tok, EulerDiscreteScheduler(), unet, None)
    pe, ne, pp, npool = pipe.encode_prompt("a young man holding a baby on his back", None, DEV, 2, True, None, None)
    ref = torch.cat([r1.hidden_states[-2][:1], r2.hidden_states[-2][:1]], dim=-1)
    assert pe.shape == (2, 77, 128 + 192) and pp.shape == (2, 128)
    assert _rel(pe[0], ref[0]) <= 2e-2 and torch.equal(pe[0], pe[1]) and _rel(pp[0], r2[0][0]) <= 2e-2
    assert ne.abs().sum() == 0 and npool.abs().sum() == 0
    pe2, ne2, _, _ = pipe.encode_prompt("a young man holding a baby on his back", None, DEV, 1, True, "two men talking", None)
    ref_n = torch.cat([r1.hidden_states[-2][1:2], r2.hidden_states[-2][1:2]], dim=-1)
    assert _rel(ne2[0], ref_n[0]) <= 2e-2
