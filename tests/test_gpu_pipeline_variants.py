"""GPU: behaviours of `DiffSenseiPipeline.__call__` the reference exposes, on the tiny config:
text-only requests (`ip_images=[]` still runs the whole IP branch on zeroed embeddings, reference :119-135),
the MLLM hand-off (`ip_image_embeds` overwrites rows 16.. of the positive embeddings only, :143-145), DDIM, re-use of
one captured plan across requests, `set_ip_scale` reaching a captured graph, determinism under a seeded generator.
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
def pipe(hip_lib):
    from transformers import CLIPVisionConfig, CLIPVisionModel, ViTMAEConfig, ViTMAEModel
    from diffsensei_amd.pipeline import DiffSenseiPipeline
    from diffsensei_amd.resampler import Resampler
    from diffsensei_amd.schedulers import EulerDiscreteScheduler
    from diffsensei_amd.unet import UNetMangaModel
    from diffsensei_amd.unet_config import random_state_dict, tiny_config
    torch.manual_seed(0)
    clip = CLIPVisionModel(CLIPVisionConfig(hidden_size=160, intermediate_size=320, num_hidden_layers=3,
                                            num_attention_heads=2, image_size=224, patch_size=14, hidden_act="quick_gelu")).eval()
    mae = ViTMAEModel(ViTMAEConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                                   image_size=224, patch_size=16, mask_ratio=0.0)).eval()
    cfg = tiny_config()
    sd = {k: v.half() for k, v in random_state_dict(cfg, 4).items()}
    unet = UNetMangaModel(cfg, device=DEV)
    unet.load_state_dict(sd)
    unet.set_manga_modules()          # installs the processors; IP K/V re-initialised from the text K/V like the reference
    sd = {k: v.float().cpu() for k, v in unet.state_dict().items()}
    rs = Resampler(dim=128, depth=2, dim_head=64, heads=2, num_queries=16, num_dummy_tokens=16, embedding_dim=160,
                   magi_embedding_dim=128, output_dim=cfg.cross_attention_dim, ff_mult=4, device=DEV).init_random(5)
    p = DiffSenseiPipeline(None, None, None, None, None, EulerDiscreteScheduler(), unet, clip)
    p.register_manga_modules(magi_image_encoder=mae, image_proj_model=rs)
    g = torch.Generator().manual_seed(9)
    common = dict(prompt="a manga panel", height=128, width=128, num_inference_steps=3, guidance_scale=7.5,
                  prompt_embeds=torch.randn(1, 77, cfg.cross_attention_dim, generator=g).half(),
                  pooled_prompt_embeds=torch.randn(1, 128, generator=g).half(), output_type="latent")
    return p, cfg, sd, rs, clip, mae, common


def _oracle(cfg, sd, rs, clip, mae, common, imgs, ip_bbox, dialog, ns, lat0, ip_embeds=None, ip_scale=0.6, ddim=False):
    from PIL import Image
    from transformers import CLIPImageProcessor, ViTImageProcessor
    from oracle.pipeline_ref import sample_loop
    from oracle.resampler_ref import resampler_forward
    from oracle.scheduler_ref import DDIMOracle, EulerDiscreteOracle
    from oracle.unet_ref import UNetOracle
    n_real = len(imgs)
    padded = list(imgs) + [Image.new("RGB", (224, 224))] * (4 - n_real)
    with torch.no_grad():
        ce = clip(CLIPImageProcessor()(images=padded, return_tensors="pt").pixel_values,
                  output_hidden_states=True).hidden_states[-2].unsqueeze(0)
        me = mae(ViTImageProcessor()(images=padded, return_tensors="pt").pixel_values).last_hidden_state[:, 0].unsqueeze(0)
        ce[0, n_real:], me[0, n_real:] = 0, 0
        rsd = {k: v.float().cpu() for k, v in rs.state_dict().items()}
        img = hq(resampler_forward(rsd, ce, me, 2, 64))
        neg = hq(resampler_forward(rsd, torch.zeros_like(ce), torch.zeros_like(me), 2, 64))
        if ip_embeds is not None:
            k = ip_embeds.shape[0]
            img[0, 16:(1 + k) * 16] = ip_embeds.float().reshape(-1, img.shape[-1])
        pe, pooled = common["prompt_embeds"].float(), common["pooled_prompt_embeds"].float()
        enc = torch.cat([torch.cat([torch.zeros_like(pe).repeat(ns, 1, 1), pe.repeat(ns, 1, 1)]),
                         torch.cat([neg.repeat(ns, 1, 1), img.repeat(ns, 1, 1)])], dim=1)
        te = torch.cat([torch.zeros(ns, pooled.shape[1]), pooled.repeat(ns, 1)])
        tid = torch.tensor([[128, 128, 0, 0, 128, 128]] * (2 * ns), dtype=torch.float32)
        bbox = torch.zeros(2 * ns, 4, 4)
        for j, bx in enumerate(ip_bbox):
            bbox[ns:, j] = torch.tensor(bx)
        db = torch.zeros(2 * ns, 8, 4, dtype=torch.float16)
        for j, bx in enumerate(dialog):
            db[ns:, j] = torch.tensor(bx).half()
        sch = (DDIMOracle() if ddim else EulerDiscreteOracle()).set_timesteps(3)
        return sample_loop(UNetOracle(cfg, sd, q=hq), DDIMOracle() if ddim else EulerDiscreteOracle(),
                           hq(lat0.float() * sch.init_noise_sigma), hq(enc), hq(te), tid, bbox, db, 7.5, 3, ip_scale, q=hq)


def test_text_only_request_runs_the_ip_branch(pipe):
    p, cfg, sd, rs, clip, mae, common = pipe
    lat0 = torch.randn(1, 4, 16, 16, generator=torch.Generator().manual_seed(1)).half()
    out = p(ip_images=[], ip_bbox=[], dialog_bbox=[], ip_scale=0.6, latents=lat0.clone(), **common).images
    ref = _oracle(cfg, sd, rs, clip, mae, common, [], [], [], 1, lat0)
    gate("test_gpu_pipeline_variants:1 " + '_rel(out, ref)', _rel(out, ref), 1.2e-2)


def test_guidance_scale_one_uses_the_negative_boxes_like_the_reference(pipe):
    """guidance_scale <= 1: the reference still hands cat([negative_ip_bbox, ip_bbox]) to a UNet batch of only the
    conditional rows, so its mask builder reads the all-zero boxes (pipeline_diffsensei.py:270-273,
    attention_processor.py:141-163); `dialog_bbox` stays the positive one.  Same here: vs the no-CFG oracle loop with
    zero boxes, and independent of the `ip_bbox` values passed."""
    from PIL import Image
    from transformers import CLIPImageProcessor, ViTImageProcessor
    from oracle.pipeline_ref import sample_loop
    from oracle.resampler_ref import resampler_forward
    from oracle.scheduler_ref import EulerDiscreteOracle
    from oracle.unet_ref import UNetOracle
    p, cfg, sd, rs, clip, mae, common = pipe
    rng = np.random.RandomState(3)
    imgs = [Image.fromarray(rng.randint(0, 256, (224, 224, 3), dtype=np.uint8)) for _ in range(2)]
    boxes = [[0.05, 0.10, 0.50, 0.95], [0.50, 0.10, 0.95, 0.95]]
    dialog = [[0.05, 0.02, 0.30, 0.15]]
    lat0 = torch.randn(1, 4, 16, 16, generator=torch.Generator().manual_seed(6)).half()
    kw = dict(common, guidance_scale=1.0)
    out = p(ip_images=list(imgs), ip_bbox=boxes, dialog_bbox=dialog, ip_scale=0.6, latents=lat0.clone(), **kw).images
    out0 = p(ip_images=list(imgs), ip_bbox=[[0.0] * 4] * 2, dialog_bbox=dialog, ip_scale=0.6, latents=lat0.clone(), **kw).images
    assert torch.equal(out, out0), "without CFG the box values must not reach the mask (reference quirk)"
    padded = list(imgs) + [Image.new("RGB", (224, 224))] * 2
    with torch.no_grad():
        ce = clip(CLIPImageProcessor()(images=padded, return_tensors="pt").pixel_values,
                  output_hidden_states=True).hidden_states[-2].unsqueeze(0)
        me = mae(ViTImageProcessor()(images=padded, return_tensors="pt").pixel_values).last_hidden_state[:, 0].unsqueeze(0)
        ce[0, 2:], me[0, 2:] = 0, 0
        img = hq(resampler_forward({k: v.float().cpu() for k, v in rs.state_dict().items()}, ce, me, 2, 64))
        enc = torch.cat([common["prompt_embeds"].float(), img], dim=1)
        te = common["pooled_prompt_embeds"].float()
        tid = torch.tensor([[128, 128, 0, 0, 128, 128]], dtype=torch.float32)
        db = torch.zeros(1, 8, 4, dtype=torch.float16)
        db[0, 0] = torch.tensor(dialog[0]).half()
        sch = EulerDiscreteOracle().set_timesteps(3)
        ref = sample_loop(UNetOracle(cfg, sd, q=hq), EulerDiscreteOracle(), hq(lat0.float() * sch.init_noise_sigma), hq(enc),
                          hq(te), tid, torch.zeros(1, 4, 4), db, 1.0, 3, 0.6, q=hq)
    gate("test_gpu_pipeline_variants:2 " + '_rel(out, ref)', _rel(out, ref), 3e-3)


def test_call_at_a_size_that_is_only_a_multiple_of_8(pipe):
    """136 x 104 px (latents 17 x 13): the reference's sliders step by 8; whole `__call__` incl. the HIP VAE to PIL."""
    from diffsensei_amd.vae import VaeConfig, VaeDecoderEngine
    p, cfg, sd, rs, clip, mae, common = pipe
    kw = {k: v for k, v in common.items() if k not in ("output_type", "height", "width")}
    kw.update(ip_images=[], ip_bbox=[], dialog_bbox=[[0.1, 0.1, 0.5, 0.4]], height=136, width=104)
    lat0 = torch.randn(1, 4, 17, 13, generator=torch.Generator().manual_seed(2)).half()
    lat = p(latents=lat0.clone(), output_type="latent", **kw).images
    assert lat.shape == (1, 4, 17, 13) and torch.isfinite(lat).all()
    assert torch.equal(lat, p(latents=lat0.clone(), output_type="latent", **kw).images)
    p.vae = VaeDecoderEngine.init_random(VaeConfig(), 3, DEV)
    try:
        pil = p(latents=lat0.clone(), **kw).images
    finally:
        p.vae = None
    assert pil[0].size == (104, 136)
    with pytest.raises(ValueError):
        p(latents=lat0.clone(), **dict(kw, height=130))          # not a multiple of 8: refused like diffusers does


def test_call_returns_images_through_the_hip_vae(pipe):
    """Whole `__call__` to images (reference :339-367): VAE decode + denormalize on the bf16 HIP decoder; "pt"/"np"/"pil"."""
    from PIL import Image
    from diffsensei_amd.vae import VaeConfig, VaeDecoderEngine
    p, cfg, sd, rs, clip, mae, common = pipe
    vae = VaeDecoderEngine.init_random(VaeConfig(), 3, DEV)
    lat0 = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(2)).half()
    kw = {k: v for k, v in common.items() if k != "output_type"}
    kw.update(ip_images=[], ip_bbox=[], dialog_bbox=[], num_samples=2)
    lat = p(latents=lat0.clone(), output_type="latent", **kw).images
    p.vae = vae
    try:
        pt = p(latents=lat0.clone(), output_type="pt", **kw).images
        npy = p(latents=lat0.clone(), output_type="np", **kw).images
        pil = p(latents=lat0.clone(), **kw).images
    finally:
        p.vae = None
    assert pt.shape == (2, 3, 128, 128) and pt.dtype == torch.float32 and 0.0 <= float(pt.min()) and float(pt.max()) <= 1.0
    assert torch.equal(pt, vae.decode(lat, return_dict=False, scaling_factor=vae.config.scaling_factor, denormalize=True)[0])
    assert npy.shape == (2, 128, 128, 3)
    assert len(pil) == 2 and isinstance(pil[0], Image.Image) and pil[0].size == (128, 128)


def test_generate_batch_equals_separate_calls(pipe):
    """Serving front-end: three different requests of one bucket in ONE UNet batch (per-request prompts, references,
    boxes, seeds, num_samples) must give what three separate `__call__`s give.  Every kernel treats batch rows
    independently; the only difference is which GEMM tiling the larger M selects (bit-identical kernels), so the
    tolerance is the fp16 noise floor."""
    from PIL import Image
    import numpy as np
    from diffsensei_amd.serving import BucketBatcher
    p, cfg, sd, rs, clip, mae, common = pipe
    rng = np.random.RandomState(3)
    img = lambda: Image.fromarray(rng.randint(0, 256, (224, 224, 3), dtype=np.uint8))
    g = torch.Generator().manual_seed(11)
    base = {k: v for k, v in common.items() if k not in ("output_type", "prompt_embeds", "pooled_prompt_embeds")}
    pe = lambda: torch.randn(1, 77, cfg.cross_attention_dim, generator=g).half()
    pool = lambda: torch.randn(1, 128, generator=g).half()
    lat = lambda n: torch.randn(n, 4, 16, 16, generator=g).half()
    reqs = [dict(base, prompt_embeds=pe(), pooled_prompt_embeds=pool(), latents=lat(1), ip_images=[img()],
                 ip_bbox=[[0.1, 0.1, 0.6, 0.9]], dialog_bbox=[[0.0, 0.0, 0.3, 0.2]], ip_scale=0.6),
            dict(base, prompt_embeds=pe(), pooled_prompt_embeds=pool(), latents=lat(2), num_samples=2, ip_images=[],
                 ip_bbox=[], dialog_bbox=[], ip_scale=0.6),
            dict(base, prompt_embeds=pe(), pooled_prompt_embeds=pool(), latents=lat(1), ip_images=[img(), img()],
                 ip_bbox=[[0.0, 0.0, 0.5, 1.0], [0.5, 0.0, 1.0, 1.0]], dialog_bbox=[], ip_scale=0.6)]
    clone = lambda r: {k: (v.clone() if torch.is_tensor(v) else (list(v) if isinstance(v, list) else v)) for k, v in r.items()}
    single = [p(output_type="latent", **clone(r)).images for r in reqs]
    b = BucketBatcher(p, max_panels=8)
    for r in reqs:
        b.submit(**clone(r))
    outs = b.run(output_type="latent")
    assert b.last_plan == [[0, 1, 2]] and p.last_run_info["batch"] == 8     # 4 panels x CFG in one plan
    for s_, o in zip(single, outs):
        assert o.shape == s_.shape
        assert _rel(o, s_) <= 2e-3, _rel(o, s_)
    with pytest.raises(ValueError):
        p.generate_batch([clone(reqs[0]), dict(clone(reqs[1]), height=256, width=256)])


def test_mllm_handoff_ip_image_embeds(pipe):
    p, cfg, sd, rs, clip, mae, common = pipe
    g = torch.Generator().manual_seed(2)
    emb = torch.randn(2, 16, cfg.cross_attention_dim, generator=g).half()
    boxes = [[0.0, 0.0, 0.5, 1.0], [0.5, 0.0, 1.0, 1.0]]
    lat0 = torch.randn(1, 4, 16, 16, generator=g).half()
    out = p(ip_images=[], ip_image_embeds=emb.to(DEV), ip_bbox=[list(b) for b in boxes], dialog_bbox=[], ip_scale=0.6,
            latents=lat0.clone(), **common).images
    ref = _oracle(cfg, sd, rs, clip, mae, common, [], boxes, [], 1, lat0, ip_embeds=emb)
    gate("test_gpu_pipeline_variants:3 " + '_rel(out, ref)', _rel(out, ref), 1.2e-2)
    plain = p(ip_images=[], ip_bbox=[], dialog_bbox=[], ip_scale=0.6, latents=lat0.clone(), **common).images
    assert _rel(out, plain) > 1e-3          # the supplied character tokens matter


def test_ddim_and_plan_reuse_and_ip_scale(pipe):
    from PIL import Image
    from diffsensei_amd.schedulers import DDIMScheduler, EulerDiscreteScheduler
    p, cfg, sd, rs, clip, mae, common = pipe
    rng = np.random.RandomState(3)
    imgs = [Image.fromarray(rng.randint(0, 256, (224, 224, 3), dtype=np.uint8))]
    boxes, dialog = [[0.1, 0.1, 0.9, 0.9]], [[0.0, 0.0, 0.4, 0.2]]
    lat0 = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(4)).half()
    kw = dict(ip_images=list(imgs), ip_bbox=[list(b) for b in boxes], dialog_bbox=[list(b) for b in dialog], num_samples=2)
    p.scheduler = DDIMScheduler()
    out_d = p(ip_scale=0.6, latents=lat0.clone(), **kw, **common).images
    ref_d = _oracle(cfg, sd, rs, clip, mae, common, imgs, boxes, dialog, 2, lat0, ddim=True)
    gate("test_gpu_pipeline_variants:4 " + '_rel(out_d, ref_d)', _rel(out_d, ref_d), 1.2e-2)
    p.scheduler = EulerDiscreteScheduler()
    a = p(ip_scale=0.6, latents=lat0.clone(), **kw, **common).images
    b = p(ip_scale=0.0, latents=lat0.clone(), **kw, **common).images      # same captured graph, new device scalar
    c = p(ip_scale=0.6, latents=lat0.clone(), **kw, **common).images
    assert torch.equal(a, c) and _rel(a, b) > 1e-3
    ref_b = _oracle(cfg, sd, rs, clip, mae, common, imgs, boxes, dialog, 2, lat0, ip_scale=0.0)
    gate("test_gpu_pipeline_variants:5 " + '_rel(b, ref_b)', _rel(b, ref_b), 1.2e-2)
    # seeded generator -> reproducible latents, like the reference's only determinism knob
    g1 = p(ip_scale=0.6, generator=torch.Generator().manual_seed(7), **kw, **common).images
    g2 = p(ip_scale=0.6, generator=torch.Generator().manual_seed(7), **kw, **common).images
    g3 = p(ip_scale=0.6, generator=torch.Generator().manual_seed(8), **kw, **common).images
    assert torch.equal(g1, g2) and not torch.equal(g1, g3)


def test_forward_flop_inventory_matches_survey(hip_lib):
    """Algorithmic work of one SDXL forward (B=2, 1024^2) from the launch plan vs SURVEY.md §8d's 13.71 TFLOP:
    the plan carries 13.71 minus the text/IP K,V projections (0.22 TF) that were hoisted out of the step loop."""
    import ctypes as C
    from diffsensei_amd import _lib
    from diffsensei_amd.engine import make_op  # noqa: F401  (table only)
    from diffsensei_amd.unet import UNetMangaModel
    from diffsensei_amd.unet_config import sdxl_config
    m = UNetMangaModel(sdxl_config(), device=DEV).init_random(0)
    eng = m.engine(2, 128, 128, 1.0)
    lib = _lib.load()
    from collections import Counter
    fl, by, name, tot, names = C.c_double(), C.c_double(), C.create_string_buffer(64), 0.0, Counter()
    for op in eng.forward_ops:
        lib.ds_op_describe(C.byref(op), name, 64, C.byref(fl), C.byref(by))
        tot += fl.value
        names[name.value.decode()] += 1
    assert abs(tot / 1e12 - (13.71 - 0.22)) < 0.05, tot / 1e12
    # the reference's call shape runs the kernels built for it (round 6): the 60 GEGLU projections of the 1280-channel level on
    # gemm_g320_kernel (256 x 320 tiles), its 60 q|k projections - and the 640-channel level's N = 640 projections - on the
    # 128 x 160 tiles of gemm_t160_kernel, the other N = 1280 projections on its 64 x 160 tiles: the UNet-vs-oracle gates of
    # tests/test_gpu_unet.py at this shape therefore cover them
    assert names["gemm_g320_kernel"] == 60 and names["gemm_t160_kernel<128 rows>"] >= 60 and names["gemm_t160_kernel"] >= 240, names
    # 963 launches with every LayerNorm a launch of its own; at this batch all 210 are folded into the GEMMs around them (the
    # 128-wide kernels' fused epilogues) at the price of 10 finalize launches (the GEGLU projections of the 64 x 64-token level are
    # gemm_pp_kernel consumers): 763, + CFG/scheduler step + counter advance = 765 launches per denoise step
    assert len(eng.forward_ops) == 763 and eng.ln_fused_launches == 210 and eng.ln_finalize_launches == 10


def test_mllm_prepass_end_to_end(pipe):
    """scripts/demo/gradio.py:85-129 as one chain: references -> Resampler tokens -> input resampler -> LLaMA greedy
    decode (forced 64-token image block) -> output resampler -> blend -> `ip_image_embeds` of the sampler; every stage
    on the HIP kernels, compared with the same chain through the fp32 oracles."""
    from PIL import Image
    from transformers import CLIPImageProcessor, ViTImageProcessor
    from diffsensei_amd.mllm import ContinuousLVLM, LlamaConfig, LlamaDecodeEngine, QwenResampler, mllm_prepass
    from oracle import llama_ref as R
    from oracle import make_golden_mllm as G
    from oracle.resampler_ref import resampler_forward
    p, cfg, sd, rs, clip, mae, common = pipe
    X = cfg.cross_attention_dim
    n_img = cfg.max_num_ips * cfg.num_vision_tokens                                        # 64
    chain = [520] + [521 + i for i in range(n_img)] + [521 + n_img]
    gi = torch.Generator().manual_seed(21)
    ids = [1] + torch.randint(3, 500, (5,), generator=gi).tolist() + chain + torch.randint(3, 500, (3,), generator=gi).tolist() + [520]
    input_ids = torch.tensor(ids)
    mask = torch.zeros(len(ids), dtype=torch.bool)
    mask[7:7 + n_img] = True
    res_in_cfg = dict(grid_size=8, embed_dim=256, num_heads=4, kv_dim=X)
    res_out_cfg = dict(grid_size=8, embed_dim=X, num_heads=4, kv_dim=256)
    llm_sd, sd_in, sd_out = G.tiny_weights(), G.tiny_resampler(res_in_cfg, 31), G.tiny_resampler(res_out_cfg, 32)
    lcfg = LlamaConfig(vocab_size=G.TINY["vocab_size"], hidden_size=256, intermediate_size=G.TINY["intermediate_size"],
                       num_hidden_layers=2, num_attention_heads=2, rms_norm_eps=G.TINY["rms_norm_eps"])
    agent = ContinuousLVLM(LlamaDecodeEngine(lcfg, llm_sd, DEV, max_positions=192, max_new_tokens=80),
                           QwenResampler(sd_in, 4, DEV), QwenResampler(sd_out, 4, DEV))
    imgs = [Image.fromarray(np.random.RandomState(s).randint(0, 256, (224, 224, 3), dtype=np.uint8)) for s in (1, 2)]
    mllm_scale, max_new, eos = 0.7, n_img + 4, 2
    got = mllm_prepass(p, agent, input_ids, mask, list(imgs), mllm_scale, img_ids_list=chain, eos_token_id=eos,
                       max_new_tokens=max_new)
    assert got.shape == (cfg.max_num_ips, cfg.num_vision_tokens, X) and got.dtype == torch.float16
    # the same chain in fp32 (padded reference slots are NOT zeroed on this path, gradio.py:91-97)
    padded = list(imgs) + [Image.new("RGB", (224, 224))] * (cfg.max_num_ips - len(imgs))
    with torch.no_grad():
        ce = clip(CLIPImageProcessor()(images=padded, return_tensors="pt").pixel_values,
                  output_hidden_states=True).hidden_states[-2].unsqueeze(0)
        me = mae(ViTImageProcessor()(images=padded, return_tensors="pt").pixel_values).last_hidden_state[:, 0].unsqueeze(0)
        rsd = {k: v.float().cpu() for k, v in rs.state_dict().items()}
        toks = hq(resampler_forward(rsd, ce, me, 2, 64))[:, cfg.num_vision_tokens:]          # [1,64,X]
    ref = R.lvlm_generate(llm_sd, R.LlamaRefConfig(**G.TINY), sd_in, sd_out, (4, 4), input_ids, toks, mask, chain, eos,
                          max_new, n_img)
    assert ref["output_ids"][:n_img + 1].tolist() == chain[1:], "oracle: forced image block"
    want = R.blend_ip_embeds(ref["img_gen_feat"], toks, mllm_scale, cfg.max_num_ips, cfg.num_vision_tokens)
    gate("test_gpu_pipeline_variants:6 " + '_rel(got, want)', _rel(got, want), 5e-3)
    # ... and into the sampler exactly like gradio.py:112-129 (`ip_images=[]`, `ip_image_embeds=`)
    boxes = [[0.0, 0.0, 0.5, 1.0], [0.5, 0.0, 1.0, 1.0], [0.0] * 4, [0.0] * 4]    # gradio.py:85-89 pads the boxes to 4
    lat0 = torch.randn(1, 4, 16, 16, generator=torch.Generator().manual_seed(4)).half()
    out = p(ip_images=[], ip_image_embeds=got, ip_bbox=[list(b) for b in boxes], dialog_bbox=[], ip_scale=0.6,
            latents=lat0.clone(), **common).images
    ref_lat = _oracle(cfg, sd, rs, clip, mae, common, [], boxes, [], 1, lat0, ip_embeds=hq(want))
    gate("test_gpu_pipeline_variants:7 " + '_rel(out, ref_lat)', _rel(out, ref_lat), 1.2e-2)


def test_callback_may_return_replaced_latents(pipe):
    """diffusers' `callback_on_step_end` contract [3P] (`latents = callback_outputs.pop("latents", latents)`): a hook that
    RETURNS new latents must have the same effect as one that edits the tensor it was handed in place; hooks that
    return nothing / the dict they got change nothing; unknown `callback_on_step_end_tensor_inputs` are refused."""
    p, cfg, sd, rs, clip, mae, common = pipe
    lat0 = torch.randn(1, 4, 16, 16, generator=torch.Generator().manual_seed(21))
    kw = dict(common, ip_images=[], ip_bbox=[], latents=lat0)
    base = p(**kw).images.clone()

    def in_place(pp, i, t, d):
        if i == 0:
            d["latents"].mul_(0.5)

    def returned(pp, i, t, d):
        return {"latents": d["latents"] * 0.5} if i == 0 else {}

    def passthrough(pp, i, t, d):
        return d

    a = p(callback_on_step_end=in_place, **kw).images.clone()
    b = p(callback_on_step_end=returned, callback_on_step_end_tensor_inputs=["latents"], **kw).images.clone()
    c = p(callback_on_step_end=passthrough, **kw).images.clone()
    assert torch.equal(a, b) and not torch.equal(a, base) and torch.equal(c, base)
    with pytest.raises(ValueError):
        p(callback_on_step_end=passthrough, callback_on_step_end_tensor_inputs=["prompt_embeds"], **kw)
