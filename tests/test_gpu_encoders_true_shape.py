"""GPU: the character and prompt encoders at their TRUE shapes and depths against `transformers` in fp32 on the CPU.

reference src/pipelines/pipeline_diffsensei.py:125-128 takes CLIP ViT-H/14's `hidden_states[-2]` (after 31 of 32 residual
layers, width 1280, 16 heads of dim 80, MLP 5120) and the ViT-MAE base CLS state (12 layers, width 768); :237-245 runs
SDXL's CLIP ViT-L/14 text encoder (12 x 768, quick_gelu) and OpenCLIP bigG/14 text encoder (32 x 1280, 20 heads, gelu,
projection 1280).  The toy-width tests in test_gpu_pipeline.py cannot show how the fp16 engines drift with depth; these do.
Weights: seeded random (no checkpoints offline), rounded to fp16 on BOTH sides so the comparison is arithmetic only.
Tolerance: relative L2 <= 2e-2 (fp16 storage between ops vs fp32), stated per assertion.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-6)).item()


def _fp16_weights(m):
    m = m.eval()
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(p.half().float())
    return m


def test_clip_vit_h14_penultimate_true_shape(hip_lib):
    from transformers import CLIPVisionConfig, CLIPVisionModel
    from diffsensei_amd.encoders import ClipVisionEngine
    torch.manual_seed(0)
    cfg = CLIPVisionConfig(hidden_size=1280, intermediate_size=5120, num_hidden_layers=32, num_attention_heads=16,
                           image_size=224, patch_size=14, hidden_act="gelu", projection_dim=1024)
    m = _fp16_weights(CLIPVisionModel(cfg))
    px = torch.randn(4, 3, 224, 224, generator=torch.Generator().manual_seed(1))
    px[3] = px[3] * 0 - 1.5                                  # a constant (black-like) reference as the padding produces
    with torch.no_grad():
        out = m(px, output_hidden_states=True)
    ref = out.hidden_states[-2]
    eng = ClipVisionEngine.from_transformers(m, DEV)
    got = eng.penultimate_hidden(px)
    assert got.shape == ref.shape == (4, 257, 1280)
    e = _rel(got, ref)
    per_img = [_rel(got[i], ref[i]) for i in range(4)]
    print(f"CLIP ViT-H/14 hidden_states[-2] (31 layers): rel-L2 {e:.3e}, per image {['%.2e' % v for v in per_img]}")
    assert e <= 2e-2 and max(per_img) <= 2e-2, (e, per_img)


def test_vit_mae_base_cls_true_shape(hip_lib):
    from transformers import ViTMAEConfig, ViTMAEModel
    from diffsensei_amd.encoders import ViTMAEEngine
    torch.manual_seed(0)
    cfg = ViTMAEConfig(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                       image_size=224, patch_size=16, mask_ratio=0.0)
    m = _fp16_weights(ViTMAEModel(cfg))
    px = torch.randn(4, 3, 224, 224, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        ref = m(px).last_hidden_state[:, 0]
    got = ViTMAEEngine.from_transformers(m, DEV).cls_embedding(px)
    assert got.shape == ref.shape == (4, 768)
    e = _rel(got, ref)
    print(f"ViT-MAE base CLS (12 layers): rel-L2 {e:.3e}")
    assert e <= 2e-2, e


@pytest.mark.parametrize("which", ["clip_l", "bigg"])
def test_sdxl_text_encoders_true_shape(hip_lib, which):
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection
    from diffsensei_amd.encoders import ClipTextEngine
    torch.manual_seed(0)
    if which == "clip_l":
        cfg = CLIPTextConfig(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                             num_attention_heads=12, max_position_embeddings=77, hidden_act="quick_gelu")
        m = _fp16_weights(CLIPTextModel(cfg))
    else:
        cfg = CLIPTextConfig(vocab_size=49408, hidden_size=1280, intermediate_size=5120, num_hidden_layers=32,
                             num_attention_heads=20, max_position_embeddings=77, hidden_act="gelu", projection_dim=1280)
        m = _fp16_weights(CLIPTextModelWithProjection(cfg))
    g = torch.Generator().manual_seed(4)
    ids = torch.full((2, 77), 49407, dtype=torch.long)
    for r, n in enumerate((14, 40)):                          # BOS, n words, EOS, EOS padding (CLIP's convention)
        ids[r, 0] = 49406
        ids[r, 1:1 + n] = torch.randint(1, 49000, (n,), generator=g)
    with torch.no_grad():
        out = m(ids, output_hidden_states=True)
    hidden, second = ClipTextEngine.from_transformers(m, DEV).encode(ids)
    e_h, e_2 = _rel(hidden, out.hidden_states[-2]), _rel(second, out[0])
    print(f"{which}: hidden_states[-2] rel-L2 {e_h:.3e}, out[0] ({'text_embeds' if which == 'bigg' else 'last_hidden_state'}) {e_2:.3e}")
    assert hidden.shape == out.hidden_states[-2].shape and second.shape == out[0].shape
    assert e_h <= 2e-2 and e_2 <= 2e-2, (e_h, e_2)
