"""Pins oracle/llama_ref.py to tests/golden/mllm_tiny.npz (transformers LlamaForCausalLM.generate + the reference's own
logits processor and QwenResampler, see oracle/make_golden_mllm.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import llama_ref as R
from oracle import make_golden_mllm as G

GOLD = os.path.join(os.path.dirname(__file__), "golden", "mllm_tiny.npz")


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(GOLD))


@pytest.fixture(scope="module")
def tiny():
    cfg = R.LlamaRefConfig(**G.TINY)
    return cfg, G.tiny_weights(), G.tiny_resampler(G.RES_IN, 11), G.tiny_resampler(G.RES_OUT, 12)


def test_qwen_resampler_matches_reference(gold, tiny):
    _, _, sd_in, sd_out = tiny
    _, _, image_embeds = G.tiny_prompt()
    got = R.qwen_resampler(sd_in, image_embeds, G.RES_IN["num_heads"])[0]
    assert np.allclose(got.numpy(), gold["input_resampler_out"], atol=2e-5)
    feats = torch.from_numpy(gold["a_hidden"][:G.N_IMG])[None]
    got = R.qwen_resampler(sd_out, feats, G.RES_OUT["num_heads"])[0]
    assert np.allclose(got.numpy(), gold["output_resampler_out"], atol=2e-5)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_generate_matches_transformers(gold, tiny, tag):
    cfg, sd, sd_in, sd_out = tiny
    input_ids, mask, image_embeds = G.tiny_prompt()
    out = R.lvlm_generate(sd, cfg, sd_in, sd_out, (G.RES_IN["num_heads"], G.RES_OUT["num_heads"]), input_ids,
                          image_embeds, mask, G.IMG_IDS, int(gold[f"{tag}_eos"]), G.MAX_NEW, G.N_IMG)
    assert out["output_ids"].tolist() == gold[f"{tag}_ids"].tolist()
    assert np.allclose(out["hidden"].numpy(), gold[f"{tag}_hidden"], atol=5e-4, rtol=1e-4)
    assert out["num_gen_imgs"] == 1 and out["ids_gen_mask"][:G.N_IMG].all() and not out["ids_gen_mask"][G.N_IMG:].any()
    assert np.allclose(out["img_gen_feat"][0].numpy(), gold["output_resampler_out"], atol=1e-4)


def test_forced_image_chain_and_zeroing():
    sc = torch.tensor([0.5, -1.0, 3.0, 0.1, -0.2, 0.3])
    chain = [1, 3, 4, 5]                                  # <img>, img_0, img_1, </img>
    out = R.image_token_processor(3, sc, chain)           # inside the chain: next id forced above the max
    assert int(out.argmax()) == 4 and float(out[4]) == pytest.approx(13.0)
    out = R.image_token_processor(0, sc, chain)           # outside: image ids (not <img>) are set to exactly 0.0
    assert out.tolist() == pytest.approx([0.5, -1.0, 3.0, 0.0, 0.0, 0.0])
    out = R.image_token_processor(5, sc, chain)           # </img> itself is not a chain member
    assert out.tolist() == pytest.approx([0.5, -1.0, 3.0, 0.0, 0.0, 0.0])


def test_blend_is_the_gradio_formula():
    a, b = torch.randn(1, 64, 8), torch.randn(1, 64, 8)
    got = R.blend_ip_embeds(a, b, 0.3, 4, 16)
    assert got.shape == (4, 16, 8) and torch.allclose(got, a.view(4, 16, 8) * 0.3 + b.view(4, 16, 8) * 0.7)
