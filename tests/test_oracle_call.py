"""CPU: the whole-`__call__` oracle (oracle/pipeline_ref.call_oracle; reference src/pipelines/pipeline_diffsensei.py:180-372)
on tiny widths - it is the checker of bench.py's `parity` object and of tests/test_gpu_call_parity.py, so its own
plumbing is pinned here: CFG layout, zeroed padded references, interrupt after `max_steps`, postprocess rounding."""
import numpy as np
import torch

from oracle.pipeline_ref import call_oracle, sample_loop
from oracle.scheduler_ref import EulerDiscreteOracle
from oracle.unet_ref import UNetOracle


class _Tok:
    model_max_length = 77

    def __call__(self, text, padding=None, max_length=77, truncation=True, return_tensors="pt"):
        words = [1 + (sum(ord(c) * (i + 1) for i, c in enumerate(w)) % 900) for w in text.split()]
        ids = [998] + words[: max_length - 2] + [999]
        ids += [999] * (max_length - len(ids))
        return type("Enc", (), {"input_ids": torch.tensor([ids])})()


def tiny_modules(seed=0):
    from transformers import (CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection, CLIPVisionConfig, CLIPVisionModel,
                              ViTMAEConfig, ViTMAEModel)
    from diffsensei_amd.unet_config import random_state_dict, tiny_config
    from diffsensei_amd.vae import VaeConfig, random_state_dict as vae_sd
    torch.manual_seed(seed)
    cfg = tiny_config()
    x = cfg.cross_attention_dim
    c1 = CLIPTextConfig(vocab_size=1000, hidden_size=x // 2, intermediate_size=x, num_hidden_layers=2, num_attention_heads=2,
                        max_position_embeddings=77, hidden_act="quick_gelu", eos_token_id=999, bos_token_id=998, pad_token_id=0)
    pooled = cfg.projection_class_embeddings_input_dim - 6 * cfg.addition_time_embed_dim
    c2 = CLIPTextConfig(vocab_size=1000, hidden_size=x // 2, intermediate_size=x, num_hidden_layers=2, num_attention_heads=2,
                        max_position_embeddings=77, hidden_act="gelu", projection_dim=pooled, eos_token_id=999,
                        bos_token_id=998, pad_token_id=0)
    clip = CLIPVisionModel(CLIPVisionConfig(hidden_size=160, intermediate_size=320, num_hidden_layers=3,
                                            num_attention_heads=2, image_size=224, patch_size=14, hidden_act="gelu")).eval()
    mae = ViTMAEModel(ViTMAEConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                                   image_size=224, patch_size=16, mask_ratio=0.0)).eval()
    vcfg = VaeConfig(block_out_channels=(32, 32, 64, 64))
    return cfg, {"tokenizer": _Tok(), "tokenizer_2": _Tok(), "text_encoder": CLIPTextModel(c1).eval(),
                 "text_encoder_2": CLIPTextModelWithProjection(c2).eval(), "image_encoder": clip, "magi": mae,
                 "resampler_sd": _resampler_sd(160, 128, 128, x, seed), "resampler_heads": 2, "resampler_dim_head": 64,
                 "vae_sd": vae_sd(vcfg, seed),
                 "vae_cfg": {"layers_per_block": vcfg.layers_per_block, "norm_num_groups": vcfg.norm_num_groups,
                             "eps": vcfg.eps, "scaling_factor": vcfg.scaling_factor}}, random_state_dict(cfg, seed)


def _resampler_sd(embedding_dim, magi_dim, dim, out_dim, seed, depth=1, n_q=16, n_dummy=16, heads=2, dim_head=64):
    g = torch.Generator().manual_seed(seed + 5)
    r = lambda *s: torch.randn(*s, generator=g) / (s[-1] ** 0.5)
    inner = heads * dim_head
    sd = {"latents": r(1, n_q, dim), "dummy_tokens": r(n_dummy, out_dim), "proj_in.weight": r(dim, embedding_dim),
          "proj_in.bias": torch.zeros(dim), "proj_in_magi.weight": r(dim, magi_dim), "proj_in_magi.bias": torch.zeros(dim),
          "proj_out.weight": r(out_dim, dim), "proj_out.bias": torch.zeros(out_dim), "norm_out.weight": torch.ones(out_dim),
          "norm_out.bias": torch.zeros(out_dim)}
    for i in range(depth):
        a, f = f"layers.{i}.0.", f"layers.{i}.1."
        sd.update({a + "norm1.weight": torch.ones(dim), a + "norm1.bias": torch.zeros(dim), a + "norm2.weight": torch.ones(dim),
                   a + "norm2.bias": torch.zeros(dim), a + "to_q.weight": r(inner, dim), a + "to_kv.weight": r(2 * inner, dim),
                   a + "to_out.weight": r(dim, inner), f + "0.weight": torch.ones(dim), f + "0.bias": torch.zeros(dim),
                   f + "1.weight": r(4 * dim, dim), f + "3.weight": r(dim, 4 * dim)})
    return sd


def test_call_oracle_layout_interrupt_and_rounding():
    cfg, mods, sd = tiny_modules()
    unet = UNetOracle(cfg, sd)
    lat0 = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(3))
    kw = dict(prompt="a manga panel of two men", negative_prompt="lowres", height=64, width=64, num_inference_steps=4,
              guidance_scale=7.5, latents=lat0, ip_scale=0.6, num_samples=2)
    tm = {}
    full = call_oracle(mods, unet, max_steps=None, timings=tm, **kw)
    cut = call_oracle(mods, unet, max_steps=2, **kw)
    assert full["steps_done"] == 4 and cut["steps_done"] == 2 and tm["steps_done"] == 4
    assert full["u8"].shape == (2, 64, 64, 3) and full["u8"].dtype == np.uint8 and full["image"].shape == (2, 3, 64, 64)
    assert not torch.allclose(full["latents"], cut["latents"])
    c = cut["conditioning"]
    n_txt, n_ip = cfg.num_text_tokens, cfg.num_ip_tokens
    assert c["enc"].shape == (4, n_txt + n_ip, cfg.cross_attention_dim) and c["bbox"].abs().sum() == 0
    assert torch.equal(c["enc"][0], c["enc"][1]) and torch.equal(c["enc"][2], c["enc"][3])          # num_samples repeat
    assert not torch.equal(c["enc"][0, :n_txt], c["enc"][2, :n_txt])                                # negative vs positive prompt
    assert torch.equal(c["enc"][0, n_txt:], c["enc"][2, n_txt:])      # text-only: all 4 references padded -> zeroed -> neg == pos
    # the interrupted loop is the plain loop cut at 2 steps
    sch = EulerDiscreteOracle().set_timesteps(4)
    ref = sample_loop(unet, EulerDiscreteOracle(), lat0 * sch.init_noise_sigma, c["enc"], c["text_embeds"], c["time_ids"],
                      c["bbox"], c["dialog"], 7.5, 4, 0.6, max_steps=2)
    assert torch.equal(ref, cut["latents"])
    # postprocess: diffusers' denormalize + numpy round-half-even to uint8
    pix = (cut["image"] / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).numpy()
    assert np.array_equal(cut["u8"], (pix * 255).round().astype("uint8"))
    # empty negative prompt -> zeros (force_zeros_for_empty_prompt), references + boxes reach the conditioning
    from PIL import Image
    rng = np.random.RandomState(0)
    img = Image.fromarray(rng.randint(0, 256, (224, 224, 3), dtype=np.uint8))
    r = call_oracle(mods, unet, "a manga panel", None, 64, 64, 4, 7.5, lat0[:1], ip_images=[img], ip_bbox=[[0.1, 0.1, 0.6, 0.9]],
                    ip_scale=0.6, dialog_bbox=[[0.0, 0.0, 0.5, 0.2]], num_samples=1, max_steps=1)["conditioning"]
    assert r["enc"][0, :n_txt].abs().sum() == 0 and r["text_embeds"][0].abs().sum() == 0
    assert not torch.equal(r["enc"][0, n_txt:], r["enc"][1, n_txt:])
    assert torch.allclose(r["bbox"][1, 0], torch.tensor([0.1, 0.1, 0.6, 0.9])) and r["bbox"][0].abs().sum() == 0
    assert r["dialog"].dtype == torch.float16 and r["dialog"][1, 0, 2] == 0.5 and r["dialog"][0].abs().sum() == 0
