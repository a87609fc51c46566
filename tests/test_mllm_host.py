"""CPU: host logic of the MLLM pre-pass mirror (diffsensei_amd/mllm.py) — configuration mapping, weight inventory,
the load-time folding of the QwenResampler constants, argument checks.  No kernel runs here (there is no CPU path:
`generate` refuses host tensors); the numerics are covered on the GPU by tests/test_gpu_mllm.py."""
import math

import pytest
import torch

from diffsensei_amd import mllm as M
from oracle import llama_ref as R
from oracle import make_golden_mllm as G


def test_config_from_transformers_and_13b_inventory():
    from transformers import LlamaConfig as HFConfig
    hf = HFConfig(vocab_size=32330, hidden_size=5120, intermediate_size=13824, num_hidden_layers=40,
                  num_attention_heads=40, rms_norm_eps=1e-5)
    cfg = M.LlamaConfig.from_hf(hf)
    assert (cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers, cfg.head_dim, cfg.kv_heads) == \
        (5120, 13824, 40, 128, 40)
    assert cfg.rms_norm_eps == pytest.approx(1e-5) and cfg.rope_theta == pytest.approx(10000.0)
    shapes = M.llama_param_shapes(cfg)
    n = sum(math.prod(s) for s in shapes.values())
    assert len(shapes) == 3 + 9 * 40 and 13.0e9 < n < 13.1e9            # LLaMA-2-13B + the added token rows
    per_token = 2 * (n - math.prod(shapes["model.embed_tokens.weight"]) - (2 * 40 + 1) * 5120)
    assert per_token == pytest.approx(25.7e9, rel=0.01)                 # the algorithmic bytes the bench prices


def test_sincos_table_is_the_reference_table():
    # oracle/make_golden_mllm.py asserts G.sincos_2d == the reference's numpy get_2d_sincos_pos_embed when it runs
    for dim, grid in ((256, 4), (128, 8), (5120, 8)):
        assert torch.equal(M.sincos_pos_embed_2d(dim, grid), G.sincos_2d(dim, grid))


def test_qwen_resampler_folding_matches_the_module_math():
    """q = (ln_q(query)+pos) Wq^T + bq and the key addend pos Wk^T + bk are weight-only: folded once at load."""
    sd = G.tiny_resampler(G.RES_IN, 11)
    rs = M.QwenResampler(sd, G.RES_IN["num_heads"], "cpu")
    E = G.RES_IN["embed_dim"]
    wi, bi = sd["attn.in_proj_weight"], sd["attn.in_proj_bias"]
    q = torch.nn.functional.layer_norm(sd["query"], (E,), sd["ln_q.weight"], sd["ln_q.bias"]) + sd["pos_embed"]
    assert torch.allclose(rs.q[0].float(), q @ wi[:E].T + bi[:E], atol=2e-3)
    add = rs._kv_addend(16)
    assert add.shape == (16, 2 * E)
    assert torch.allclose(add[:, :E].float(), sd["pos_embed"] @ wi[E:2 * E].T + bi[E:2 * E], atol=2e-3)
    assert torch.allclose(add[:, E:].float(), bi[2 * E:][None].expand(16, -1), atol=1e-3)
    # a different token count interpolates the table like get_abs_pos (qwen_resampler.py:15-33): 4x4 -> 2x2
    add4 = rs._kv_addend(4)
    pos4 = torch.nn.functional.interpolate(sd["pos_embed"].reshape(1, 4, 4, -1).permute(0, 3, 1, 2), size=(2, 2),
                                           mode="bicubic", align_corners=False).permute(0, 2, 3, 1).flatten(0, 2)
    assert torch.allclose(add4[:, :E].float(), pos4 @ wi[E:2 * E].T + bi[E:2 * E], atol=2e-3)
    # and the folded form is the module: oracle(x) == attention over (x Wkv^T + addend) with the constant q
    x = torch.randn(1, 16, G.RES_IN["kv_dim"], generator=torch.Generator().manual_seed(0))
    xn = torch.nn.functional.layer_norm(x[0] @ sd["kv_proj.weight"].T, (E,), sd["ln_kv.weight"], sd["ln_kv.bias"])
    kv = xn @ wi[E:].T + add.float()
    h, d = G.RES_IN["num_heads"], E // G.RES_IN["num_heads"]
    s = torch.einsum("qhd,lhd->hql", rs.q[0].float().view(16, h, d), kv[:, :E].view(16, h, d)) / math.sqrt(d)
    o = torch.einsum("hql,lhd->qhd", s.softmax(-1), kv[:, E:].view(16, h, d)).reshape(16, E)
    o = o @ sd["attn.out_proj.weight"].T + sd["attn.out_proj.bias"]
    assert torch.allclose(o, R.qwen_resampler(sd, x, h)[0], atol=5e-3)


def test_generate_argument_contract():
    agent = M.ContinuousLVLM(None, None, None)
    with pytest.raises(NotImplementedError):
        agent.generate(input_ids=torch.tensor([[1, 2]]), num_beams=4, img_ids_list=[5, 6, 7], eos_token_id=2)
    with pytest.raises(NotImplementedError):
        agent.generate(input_ids=torch.tensor([[1, 2]]), logits_processor=[object()], img_ids_list=[5, 6, 7],
                       eos_token_id=2)
    with pytest.raises(ValueError):
        agent.generate(input_ids=torch.tensor([[1, 2]]))                 # neither tokenizer nor img_ids_list
    with pytest.raises(ValueError):
        agent.generate(input_ids=torch.tensor([[1, 2]]), img_ids_list=[5, 6, 7])   # no eos id


def test_engine_has_no_cpu_path():
    cfg = M.LlamaConfig(vocab_size=64, hidden_size=128, intermediate_size=256, num_hidden_layers=1,
                        num_attention_heads=1)
    with pytest.raises(ValueError):
        M.LlamaDecodeEngine(cfg, {}, "cpu", prompt_path="torch")
    with pytest.raises(ValueError):
        M.LlamaDecodeEngine(M.LlamaConfig(vocab_size=64, hidden_size=96, intermediate_size=256, num_hidden_layers=1,
                                          num_attention_heads=1), {}, "cpu")             # head_dim 96: no kernel
    g = torch.Generator().manual_seed(0)
    sd = {k: torch.randn(s, generator=g) * 0.05 for k, s in M.llama_param_shapes(cfg).items()}
    eng = M.LlamaDecodeEngine(cfg, sd, "cpu", max_positions=32, max_new_tokens=8)        # packing is host work
    assert eng.wqkv[0].shape == (3 * 128, 128) and eng.wgu[0].shape == (512, 128)
    stacked = torch.cat([sd["model.layers.0.self_attn.%s_proj.weight" % n] for n in "qkv"])
    assert torch.equal(eng.wqkv[0], stacked.half())                  # stacked, NOT folded: the RMSNorm gains stay vectors
    assert torch.equal(eng.g_in[0], sd["model.layers.0.input_layernorm.weight"].half())
    assert torch.equal(eng.g_post[0], sd["model.layers.0.post_attention_layernorm.weight"].half())
    assert len(eng.tensors()) == 3 + 6 * cfg.num_hidden_layers
    assert eng.weight_bytes_per_token() == 2 * (3 * 128 * 128 + 128 * 128 + 512 * 128 + 128 * 256 + 64 * 128 + 128)
    with pytest.raises(ValueError):
        eng.generate(torch.zeros(4, 128, dtype=torch.float16), 1, 2, 4)                  # host tensor: refused


class _SentencePieceLikeTokenizer:
    """LLaMA-style: every encode() result starts with the '▁' id (29871), like the tokenizer the reference uses."""
    eos_token_id = 2

    def __init__(self, boi=32100, n=64):
        self.vocab = {"<img>": boi, "</img>": boi + n + 1}
        self.vocab.update({f"<img_{i:05d}>": boi + 1 + i for i in range(n)})

    def encode(self, s, add_special_tokens=False):
        ids, i = [29871], 0
        while i < len(s):
            j = s.index(">", i) + 1
            ids.append(self.vocab[s[i:j]])
            i = j
        return ids


def test_image_token_ids_follow_the_reference_tokenizer_convention():
    """reference seed_x.py:139-141 / gradio.py:44-45 index past the sentencepiece prefix id ([1], [1:]) while the logits
    processor keeps the whole encoded list (generation.py:15-17)."""
    tok = _SentencePieceLikeTokenizer()
    n = 64
    # what the reference computes
    ref_eoi = tok.encode(M.EOI_TOKEN, add_special_tokens=False)[1]
    ref_img = tok.encode("".join(M.IMG_TOKEN.format(i) for i in range(n)), add_special_tokens=False)[1:]
    ref_chain = tok.encode("".join([M.BOI_TOKEN] + [M.IMG_TOKEN.format(i) for i in range(n)] + [M.EOI_TOKEN]),
                           add_special_tokens=False)
    chain, eoi, img = M.image_token_ids(tok, n)
    assert chain == ref_chain and chain[0] == 29871 and len(chain) == n + 3
    assert eoi == ref_eoi and img == ref_img and len(img) == n
    # an explicit list without the prefix (what the golden fixtures pass) gives the same ids
    chain2, eoi2, img2 = M.image_token_ids(None, n, ref_chain[1:])
    assert (eoi2, img2) == (eoi, img) and chain2 == ref_chain[1:]
    with pytest.raises(ValueError):
        M.image_token_ids(None, n, ref_chain[:10])
    with pytest.raises(ValueError):
        M.image_token_ids(None, n)
