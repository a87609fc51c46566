"""CPU: the VAE decoder's host side (parameter inventory in diffusers' key names) and the fp32 oracle's structure."""
import torch

from diffsensei_amd.vae import VaeConfig, random_state_dict, vae_param_shapes


def test_param_inventory_matches_diffusers_decoder_layout():
    sh = vae_param_shapes(VaeConfig())
    assert len(sh) == 140
    assert sh["decoder.conv_in.weight"] == (512, 4, 3, 3) and sh["decoder.conv_out.weight"] == (3, 128, 3, 3)
    assert sh["decoder.up_blocks.2.resnets.0.conv_shortcut.weight"] == (256, 512, 1, 1)
    assert sh["decoder.up_blocks.3.resnets.0.conv_shortcut.weight"] == (128, 256, 1, 1)
    assert "decoder.up_blocks.3.upsamplers.0.conv.weight" not in sh and "decoder.up_blocks.2.upsamplers.0.conv.weight" in sh
    assert sh["decoder.mid_block.attentions.0.to_out.0.weight"] == (512, 512)
    n = sum(int(torch.tensor(s).prod()) for s in sh.values())
    # public anchor: the SD/SDXL AutoencoderKL has 83 653 863 parameters = encoder 34 163 592 + decoder 49 490 179 +
    # quant_conv 72 + post_quant_conv 20; this inventory is the decoder + post_quant_conv
    dec = sum(int(torch.tensor(s).prod()) for k, s in sh.items() if k.startswith("decoder."))
    assert dec == 49_490_179 and n == 49_490_179 + 20 and 34_163_592 + dec + 72 + 20 == 83_653_863


def test_oracle_decode_structure():
    from oracle.vae_ref import vae_decode
    cfg = VaeConfig(block_out_channels=(32, 32, 64, 64), layers_per_block=1, norm_num_groups=8)
    sd = random_state_dict(cfg, 0)
    z = torch.randn(2, 4, 6, 10, generator=torch.Generator().manual_seed(1))
    img = vae_decode(sd, z, cfg.layers_per_block, cfg.norm_num_groups, cfg.eps)
    assert img.shape == (2, 3, 48, 80) and torch.isfinite(img).all()
    # batch items are independent, and conv_out's bias is a plain per-channel offset
    assert torch.allclose(vae_decode(sd, z[1:], 1, 8, cfg.eps), img[1:], atol=1e-5)
    sd2 = dict(sd)
    sd2["decoder.conv_out.bias"] = sd["decoder.conv_out.bias"] + torch.tensor([1.0, -2.0, 0.5])
    assert torch.allclose(vae_decode(sd2, z, 1, 8, cfg.eps) - img, torch.tensor([1.0, -2.0, 0.5]).view(1, 3, 1, 1).expand_as(img), atol=1e-5)


def test_latents_mean_std_fold_equals_the_reference_formula():
    """reference src/pipelines/pipeline_diffsensei.py:348-357: `latents * latents_std / scaling_factor + latents_mean`, then
    decode.  The product folds that per-channel affine map into post_quant_conv (`vae.fold_latents_affine`); on the oracle
    decoder the folded weights applied to `latents / scaling_factor` must give the reference formula's image."""
    from diffsensei_amd.vae import fold_latents_affine
    from oracle.vae_ref import vae_decode
    cfg = VaeConfig(block_out_channels=(32, 32, 64, 64), layers_per_block=1, norm_num_groups=8)
    sd = random_state_dict(cfg, 3)
    mean, std, sf = [0.3, -0.2, 0.05, 1.1], [1.2, 0.7, 2.0, 0.9], 0.13025
    lat = torch.randn(2, 4, 5, 7, generator=torch.Generator().manual_seed(2))
    ref = vae_decode(sd, lat * torch.tensor(std).view(1, 4, 1, 1) / sf + torch.tensor(mean).view(1, 4, 1, 1), 1, 8, cfg.eps)
    w2, b2 = fold_latents_affine(sd["post_quant_conv.weight"], sd["post_quant_conv.bias"], mean, std)
    sd2 = dict(sd)
    sd2["post_quant_conv.weight"], sd2["post_quant_conv.bias"] = w2.view(4, 4, 1, 1), b2
    got = vae_decode(sd2, lat / sf, 1, 8, cfg.eps)
    assert torch.allclose(got, ref, atol=2e-5, rtol=1e-5), (got - ref).abs().max()
    import pytest
    with pytest.raises(ValueError):          # the pair comes together
        from diffsensei_amd.vae import VaeDecoderEngine
        VaeDecoderEngine(VaeConfig(latents_mean=mean), {}, device="cpu")
