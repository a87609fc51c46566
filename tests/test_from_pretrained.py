"""`DiffSenseiPipeline.from_pretrained(dir, unet=, image_encoder=, torch_dtype=)` — the reference's construction recipe
(scripts/demo/gradio_wo_mllm.py:161-200) on a synthetic diffusers-layout directory written by the test (no checkpoint is
reachable offline): model_index.json, scheduler/scheduler_config.json, vae/, text_encoder/, text_encoder_2/, tokenizer/,
tokenizer_2/, unet/ — safetensors and .bin weights.  CPU: the directory is parsed into engines (no compute).
GPU: the loaded pipeline produces the same panel as one assembled from the same components through `__init__`."""
import json
import os

import pytest
import torch

DEV = "cuda"


def _write_tokenizer(folder):
    os.makedirs(folder, exist_ok=True)
    words = ["a", "manga", "panel", "of", "two", "kids", "b", "c"]
    vocab = {"<|startoftext|>": 0, "<|endoftext|>": 1}
    for w in words:
        vocab[w + "</w>"] = len(vocab)
    for ch in sorted(set("".join(words))):
        vocab.setdefault(ch, len(vocab))
        vocab.setdefault(ch + "</w>", len(vocab))
    json.dump(vocab, open(os.path.join(folder, "vocab.json"), "w"))
    merges = ["#version: 0.2"]
    for w in words:                      # merge every word left to right so whole words become single tokens
        parts = list(w[:-1]) + [w[-1] + "</w>"]
        while len(parts) > 1:
            merges.append(f"{parts[0]} {parts[1]}")
            parts = [parts[0] + parts[1]] + parts[2:]
            vocab.setdefault(parts[0], len(vocab))
    json.dump(vocab, open(os.path.join(folder, "vocab.json"), "w"))
    open(os.path.join(folder, "merges.txt"), "w").write("\n".join(dict.fromkeys(merges)) + "\n")
    json.dump({"model_max_length": 77, "bos_token": "<|startoftext|>", "eos_token": "<|endoftext|>",
               "unk_token": "<|endoftext|>", "pad_token": "<|endoftext|>", "tokenizer_class": "CLIPTokenizer"},
              open(os.path.join(folder, "tokenizer_config.json"), "w"))
    return len(vocab)


def make_checkpoint_dir(root, seed=0):
    """A tiny `image_generator/`-like directory in diffusers layout; returns the pieces used to write it."""
    from safetensors.torch import save_file
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection
    from diffsensei_amd.unet_config import random_state_dict, tiny_config
    from diffsensei_amd.vae import VaeConfig, random_state_dict as vae_sd
    torch.manual_seed(seed)
    os.makedirs(root, exist_ok=True)
    json.dump({"_class_name": "StableDiffusionXLPipeline", "_diffusers_version": "0.27.0",
               "force_zeros_for_empty_prompt": True, "scheduler": ["diffusers", "EulerDiscreteScheduler"],
               "text_encoder": ["transformers", "CLIPTextModel"],
               "text_encoder_2": ["transformers", "CLIPTextModelWithProjection"],
               "tokenizer": ["transformers", "CLIPTokenizer"], "tokenizer_2": ["transformers", "CLIPTokenizer"],
               "unet": ["diffusers", "UNet2DConditionModel"], "vae": ["diffusers", "AutoencoderKL"],
               "image_encoder": [None, None], "feature_extractor": [None, None]},
              open(os.path.join(root, "model_index.json"), "w"))
    os.makedirs(os.path.join(root, "scheduler"))
    json.dump({"_class_name": "EulerDiscreteScheduler", "_diffusers_version": "0.27.0", "beta_start": 0.00085,
               "beta_end": 0.012, "beta_schedule": "scaled_linear", "num_train_timesteps": 1000, "steps_offset": 1,
               "timestep_spacing": "leading", "prediction_type": "epsilon", "interpolation_type": "linear",
               "use_karras_sigmas": False, "trained_betas": None, "clip_sample": False, "set_alpha_to_one": False,
               "skip_prk_steps": True, "sample_max_value": 1.0},
              open(os.path.join(root, "scheduler", "scheduler_config.json"), "w"))
    vocab = max(_write_tokenizer(os.path.join(root, "tokenizer")), _write_tokenizer(os.path.join(root, "tokenizer_2")))
    # text encoders: widths 64 + 192 = the tiny UNet's cross_attention_dim 256; pooled 128 = projection_dim
    t1 = CLIPTextConfig(vocab_size=vocab, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
                        max_position_embeddings=77, hidden_act="quick_gelu", bos_token_id=0, eos_token_id=1, pad_token_id=1)
    t2 = CLIPTextConfig(vocab_size=vocab, hidden_size=192, intermediate_size=256, num_hidden_layers=2, num_attention_heads=3,
                        max_position_embeddings=77, hidden_act="gelu", projection_dim=128, bos_token_id=0, eos_token_id=1,
                        pad_token_id=1)
    te1, te2 = CLIPTextModel(t1).eval(), CLIPTextModelWithProjection(t2).eval()
    te1.save_pretrained(os.path.join(root, "text_encoder"), safe_serialization=True)        # model.safetensors
    os.makedirs(os.path.join(root, "text_encoder_2"))
    t2.to_json_file(os.path.join(root, "text_encoder_2", "config.json"))
    torch.save(te2.state_dict(), os.path.join(root, "text_encoder_2", "pytorch_model.bin"))  # .bin pickle
    vcfg = VaeConfig(block_out_channels=(128, 128, 256, 512), layers_per_block=1)
    os.makedirs(os.path.join(root, "vae"))
    json.dump({"_class_name": "AutoencoderKL", "block_out_channels": list(vcfg.block_out_channels), "layers_per_block": 1,
               "latent_channels": 4, "out_channels": 3, "norm_num_groups": 32, "scaling_factor": 0.13025,
               "force_upcast": True, "in_channels": 3, "sample_size": 128}, open(os.path.join(root, "vae", "config.json"), "w"))
    vsd = {k: v.contiguous() for k, v in vae_sd(vcfg, seed + 2).items()}
    vsd["encoder.conv_in.weight"] = torch.zeros(128, 3, 3, 3)                               # extra keys are ignored
    save_file(vsd, os.path.join(root, "vae", "diffusion_pytorch_model.safetensors"))
    ucfg = tiny_config()
    os.makedirs(os.path.join(root, "unet"))
    json.dump({"_class_name": "UNet2DConditionModel", **{k: (list(v) if isinstance(v, tuple) else v)
                                                          for k, v in ucfg.to_dict().items()}},
              open(os.path.join(root, "unet", "config.json"), "w"))
    usd = {k: v.half().contiguous() for k, v in random_state_dict(ucfg, seed + 4).items()}
    torch.save(usd, os.path.join(root, "unet", "pytorch_model.bin"))
    return {"te1": te1, "te2": te2, "vae_cfg": vcfg, "vae_sd": vsd, "unet_cfg": ucfg, "unet_sd": usd}


def test_from_pretrained_reads_a_diffusers_directory(tmp_path):
    from diffsensei_amd.encoders import ClipTextEngine
    from diffsensei_amd.pipeline import DiffSenseiPipeline
    from diffsensei_amd.schedulers import EulerDiscreteScheduler
    from diffsensei_amd.unet import UNetMangaModel
    from diffsensei_amd.vae import VaeDecoderEngine
    root = str(tmp_path / "image_generator")
    made = make_checkpoint_dir(root)
    # the recipe of gradio_wo_mllm.py:161-194: the UNet is built by the caller and handed in
    unet = UNetMangaModel.from_config(root, subfolder="unet", torch_dtype=torch.float16, device="cpu")
    unet.set_manga_modules(max_num_ips=4, num_vision_tokens=16, max_num_dialogs=8)
    unet.load_state_dict(torch.load(os.path.join(root, "unet", "pytorch_model.bin")))
    pipe = DiffSenseiPipeline.from_pretrained(root, unet=unet, image_encoder=None, torch_dtype=torch.float16)
    assert pipe.unet is unet and pipe.device == torch.device("cpu")
    assert isinstance(pipe.scheduler, EulerDiscreteScheduler) and pipe.scheduler.steps_offset == 1
    assert isinstance(pipe.vae, VaeDecoderEngine) and pipe.vae.config.block_out_channels == (128, 128, 256, 512)
    assert abs(pipe.vae.config.scaling_factor - 0.13025) < 1e-9
    assert isinstance(pipe.text_encoder, ClipTextEngine) and isinstance(pipe.text_encoder_2, ClipTextEngine)
    assert pipe.text_encoder.hidden == 64 and pipe.text_encoder_2.hidden == 192
    assert pipe.text_encoder.text_projection is None and pipe.text_encoder_2.text_projection.shape == (128, 192)
    assert len(pipe.text_encoder.layers) == 2 and pipe.text_encoder.act == "quick_gelu" and pipe.text_encoder_2.act == "gelu"
    by_suffix = lambda sd, suf: next(v for k, v in sd.items() if k.endswith(suf))   # transformers 4.x / 5.x key prefixes differ
    w = by_suffix(made["te1"].state_dict(), "embeddings.token_embedding.weight")
    assert torch.equal(pipe.text_encoder.tok_emb, w.half())                                 # safetensors component
    w2 = by_suffix(made["te2"].state_dict(), "text_projection.weight")
    assert torch.equal(pipe.text_encoder_2.text_projection, w2.half())                      # .bin component
    ids = pipe.tokenizer("a manga panel", padding="max_length", max_length=pipe.tokenizer.model_max_length,
                         truncation=True, return_tensors="pt").input_ids
    assert ids.shape == (1, 77) and ids[0, 0] == 0 and ids[0, 4] == 1 and len(set(ids[0, 1:4].tolist())) == 3
    assert pipe.force_zeros_for_empty_prompt is True and pipe.image_encoder is None
    # no `unet=`: read from unet/ (config.json + weights)
    pipe2 = DiffSenseiPipeline.from_pretrained(root, device="cpu")
    assert pipe2.unet.config.block_out_channels == (64, 128, 256) and len(pipe2.unet.state_dict()) == len(made["unet_sd"])
    k = "down_blocks.1.attentions.0.transformer_blocks.0.attn2.processor.to_k_ip.weight"
    assert torch.equal(pipe2.unet.state_dict()[k], made["unet_sd"][k])
    # errors
    with pytest.raises(FileNotFoundError):
        DiffSenseiPipeline.from_pretrained(str(tmp_path / "nope"), unet=unet)
    with pytest.raises(ValueError):
        DiffSenseiPipeline.from_pretrained(root, unet=unet, torch_dtype=torch.float32)
    os.remove(os.path.join(root, "scheduler", "scheduler_config.json"))
    json.dump({"_class_name": "DPMSolverMultistepScheduler"}, open(os.path.join(root, "scheduler", "scheduler_config.json"), "w"))
    with pytest.raises(NotImplementedError):
        DiffSenseiPipeline.from_pretrained(root, unet=unet)


def test_set_attn_processor_refuses_what_the_plan_does_not_run():
    """diffusers protocol `unet.set_attn_processor(dict)` (reference src/models/unet.py:84): only the reference's two
    processor classes are executed by the launch plan; anything else must raise instead of being silently ignored."""
    from diffsensei_amd.attention_processor import AttnProcessor2_0, MaskedIPAttnProcessor2_0
    from diffsensei_amd.unet import UNetMangaModel
    from diffsensei_amd.unet_config import tiny_config
    m = UNetMangaModel(tiny_config(), device="cpu")
    procs = dict(m.attn_processors)
    m.set_attn_processor(procs)                                        # the default AttnProcessor2_0 table is fine
    assert all(isinstance(p, AttnProcessor2_0) for p in m.attn_processors.values())

    class Custom:
        def __call__(self, *a, **k):
            raise AssertionError

    bad = dict(procs)
    bad[next(iter(bad))] = Custom()
    with pytest.raises(ValueError):
        m.set_attn_processor(bad)
    with pytest.raises(ValueError):
        m.set_attn_processor({"not.a.layer.processor": AttnProcessor2_0()})
    missing = dict(procs)
    missing.pop(next(iter(missing)))
    with pytest.raises(ValueError):
        m.set_attn_processor(missing)
    a1 = next(n for n in procs if n.endswith("attn1.processor"))
    wrong = dict(procs)
    wrong[a1] = MaskedIPAttnProcessor2_0(hidden_size=128, cross_attention_dim=256, num_ip_tokens=64, num_dummy_tokens=16,
                                         device="cpu")
    with pytest.raises(ValueError):
        m.set_attn_processor(wrong)                                    # an IP processor on a self-attention slot


@pytest.mark.gpu
def test_from_pretrained_pipeline_runs_and_equals_direct_construction(tmp_path, hip_lib):
    from transformers import CLIPVisionConfig, CLIPVisionModel, ViTMAEConfig, ViTMAEModel
    from diffsensei_amd.pipeline import DiffSenseiPipeline
    from diffsensei_amd.resampler import Resampler
    from diffsensei_amd.schedulers import EulerDiscreteScheduler
    from diffsensei_amd.unet import UNetMangaModel
    from diffsensei_amd.vae import VaeDecoderEngine
    root = str(tmp_path / "image_generator")
    made = make_checkpoint_dir(root, seed=3)
    torch.manual_seed(1)
    clip = CLIPVisionModel(CLIPVisionConfig(hidden_size=160, intermediate_size=320, num_hidden_layers=3,
                                            num_attention_heads=2, image_size=224, patch_size=14, hidden_act="quick_gelu")).eval()
    mae = ViTMAEModel(ViTMAEConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                                   image_size=224, patch_size=16, mask_ratio=0.0)).eval()

    def unet_from_dir():
        u = UNetMangaModel.from_config(root, subfolder="unet", torch_dtype=torch.float16, device=DEV)
        u.set_manga_modules(max_num_ips=4, num_vision_tokens=16, max_num_dialogs=8)
        u.load_state_dict(torch.load(os.path.join(root, "unet", "pytorch_model.bin")))
        return u

    def finish(p):
        rs = Resampler(dim=128, depth=2, dim_head=64, heads=2, num_queries=16, num_dummy_tokens=16, embedding_dim=160,
                       magi_embedding_dim=128, output_dim=256, ff_mult=4, device=DEV).init_random(5)
        p.register_manga_modules(magi_image_encoder=mae, image_proj_model=rs)
        return p.to(device=DEV, dtype=torch.float16)

    loaded = finish(DiffSenseiPipeline.from_pretrained(root, unet=unet_from_dir(), image_encoder=clip, torch_dtype=torch.float16))
    direct = finish(DiffSenseiPipeline(VaeDecoderEngine.from_state_dict(made["vae_sd"], made["vae_cfg"], DEV), made["te1"],
                                       made["te2"], loaded.tokenizer, loaded.tokenizer_2, EulerDiscreteScheduler(),
                                       unet_from_dir(), clip))
    kw = dict(prompt="a manga panel of two kids", height=128, width=128, num_inference_steps=3, guidance_scale=7.5,
              negative_prompt="b c", ip_images=[], ip_bbox=[], dialog_bbox=[[0.1, 0.1, 0.4, 0.3]], ip_scale=0.6,
              output_type="pt")
    a = loaded(generator=torch.Generator().manual_seed(0), **kw).images
    b = direct(generator=torch.Generator().manual_seed(0), **kw).images
    assert a.shape == (1, 3, 128, 128) and torch.isfinite(a).all() and a.std() > 1e-3
    assert torch.equal(a, b)
    pil = loaded(generator=torch.Generator().manual_seed(0), **dict(kw, output_type="pil")).images
    assert pil[0].size == (128, 128)
