"""GPU: `DiffSenseiPipeline.__call__` from the prompt STRING to the PIL bytes against the whole-call CPU oracle
(oracle/pipeline_ref.call_oracle: transformers text / image encoders in fp32, HF image processors, oracle Resampler,
UNet, Euler loop, fp32 VAE decode, diffusers' postprocess rounding) - reference src/pipelines/pipeline_diffsensei.py:180-372.

Same prompt, negative prompt, references, boxes, initial noise and weights on both sides; the loop is interrupted after 2 of
4 steps through the reference's own early-exit (`pipe._interrupt`, :314-315) raised from `callback_on_step_end`.
This is the pytest twin of bench.py's `parity` object (which runs the SDXL-size models on BASELINE configs[0]).
Tolerances (round 6: <= ~3-4x the measured 1.9e-3 / 2.1e-3 / no byte off by more than 1 LSB; until round 5 they were 5e-2 / 5e-2 /
2 % by more than 2 LSB - gates that gated nothing): latents relative L2 <= 8e-3, image (uint8 / 255) relative L2 <= 8e-3,
bytes off by more than 1 LSB <= 0.5 % (measured 0.004 %), none by more than 3 (measured 2).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_call_prompt_to_pil_vs_call_oracle(hip_lib):
    from PIL import Image
    from diffsensei_amd.pipeline import DiffSenseiPipeline
    from diffsensei_amd.resampler import Resampler
    from diffsensei_amd.schedulers import EulerDiscreteScheduler
    from diffsensei_amd.unet import UNetMangaModel
    from diffsensei_amd.vae import VaeConfig, VaeDecoderEngine, random_state_dict as vae_sd
    from oracle.pipeline_ref import call_oracle
    from oracle.unet_ref import UNetOracle
    from tests.test_oracle_call import tiny_modules
    cfg, mods, sd = tiny_modules(seed=1)
    for name in ("text_encoder", "text_encoder_2", "image_encoder", "magi"):      # identical (fp16-representable) weights
        with torch.no_grad():
            for p in mods[name].parameters():
                p.copy_(p.half().float())
    sd = {k: v.half() for k, v in sd.items()}
    unet = UNetMangaModel(cfg, device=DEV)
    unet.load_state_dict(sd)
    rs = Resampler(dim=128, depth=1, dim_head=64, heads=2, num_queries=16, num_dummy_tokens=16, embedding_dim=160,
                   magi_embedding_dim=128, output_dim=cfg.cross_attention_dim, ff_mult=4, device=DEV).init_random(3)
    mods["resampler_sd"] = {k: v.float().cpu() for k, v in rs.state_dict().items()}
    vcfg = VaeConfig()
    vsd = {k: v.to(torch.bfloat16).float() for k, v in vae_sd(vcfg, 4).items()}
    mods["vae_sd"] = vsd
    mods["vae_cfg"] = {"layers_per_block": vcfg.layers_per_block, "norm_num_groups": vcfg.norm_num_groups, "eps": vcfg.eps,
                       "scaling_factor": vcfg.scaling_factor}
    vae = VaeDecoderEngine.from_state_dict(vsd, vcfg, DEV)
    pipe = DiffSenseiPipeline(vae, mods["text_encoder"], mods["text_encoder_2"], mods["tokenizer"], mods["tokenizer_2"],
                              EulerDiscreteScheduler(), unet, mods["image_encoder"])
    pipe.register_manga_modules(magi_image_encoder=mods["magi"], image_proj_model=rs)
    rng = np.random.RandomState(0)
    imgs = [Image.fromarray(rng.randint(0, 256, (224, 224, 3), dtype=np.uint8)) for _ in range(2)]
    size, steps, cut, ns = 128, 4, 2, 2
    lat0 = torch.randn(ns, 4, size // 8, size // 8, generator=torch.Generator().manual_seed(9))
    req = dict(prompt="a young man holding a baby on his back", negative_prompt="lowres, bad anatomy", height=size, width=size,
               num_inference_steps=steps, guidance_scale=7.5, ip_scale=0.6,
               ip_bbox=[[0.05, 0.10, 0.50, 0.95], [0.50, 0.10, 0.95, 0.95]],
               dialog_bbox=[[0.05, 0.02, 0.30, 0.15], [0.65, 0.02, 0.95, 0.15]])
    ref = call_oracle(mods, UNetOracle(cfg, sd), latents=lat0, ip_images=imgs, num_samples=ns, max_steps=cut, **req)
    seen = []

    def stop(p, i, t, kw):
        seen.append(i)
        assert kw["latents"].is_cuda
        if i + 1 >= cut:
            p._interrupt = True
        return kw

    outs = {}
    for use_graph in (False, True):
        pipe.use_graph = use_graph
        seen.clear()
        lat = pipe(latents=lat0.clone(), ip_images=list(imgs), num_samples=ns, output_type="latent",
                   callback_on_step_end=stop, **req).images
        assert seen == list(range(cut)) and pipe.interrupt
        pil = pipe(latents=lat0.clone(), ip_images=list(imgs), num_samples=ns, output_type="pil",
                   callback_on_step_end=stop, **req).images
        outs[use_graph] = (lat.clone(), np.stack([np.asarray(im) for im in pil]))
    assert torch.equal(outs[False][0], outs[True][0]) and np.array_equal(outs[False][1], outs[True][1])
    # a later un-interrupted call runs all steps again (the flag is reset at entry like reference :226)
    full = pipe(latents=lat0.clone(), ip_images=list(imgs), num_samples=ns, output_type="latent", **req).images
    assert not pipe.interrupt and not torch.equal(full, outs[True][0])
    # ... and neither does `generate_batch` inherit the flag of an interrupted `__call__`
    pipe(latents=lat0.clone(), ip_images=list(imgs), num_samples=ns, output_type="latent", callback_on_step_end=stop, **req)
    assert pipe.interrupt
    batch = pipe.generate_batch([dict(req, latents=lat0.clone(), ip_images=list(imgs), num_samples=ns)], output_type="latent")[0]
    assert not pipe.interrupt and ((batch.float() - full.float()).norm() / full.float().norm()).item() <= 1e-3
    lat, u8 = outs[True]
    rl = ref["latents"]
    e_lat = ((lat.float().cpu() - rl).norm() / rl.norm()).item()
    a, b = u8.astype(np.float64) / 255, ref["u8"].astype(np.float64) / 255
    e_img = float(np.linalg.norm(a - b) / np.linalg.norm(b))
    d = np.abs(u8.astype(np.int16) - ref["u8"].astype(np.int16))
    print(f"__call__ vs call_oracle (tiny widths, {cut} of {steps} steps): latents rel-L2 {e_lat:.3e}, image rel-L2 {e_img:.3e}, "
          f"bytes differing {float((d != 0).mean()):.4f}, by > 1 LSB {float((d > 1).mean()):.5f}, max {int(d.max())}, "
          f"image std {ref['u8'].std():.1f}")
    assert u8.shape == ref["u8"].shape == (ns, size, size, 3)
    from tests._gates import gate
    gate("__call__ latents rel-L2 vs call_oracle", e_lat, 8e-3)
    gate("__call__ uint8 image rel-L2 vs call_oracle", e_img, 8e-3)
    gate("__call__ share of bytes off by > 1 LSB", float((d > 1).mean()), 5e-3)
    gate("__call__ largest byte difference", int(d.max()), 3)
