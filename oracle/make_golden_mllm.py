#!/usr/bin/env python
"""Generates tests/golden/mllm_tiny.npz — the parity pin of oracle/llama_ref.py (run in the build container only:
it imports the reference from /root/reference and transformers; neither exists on the GPU box).

What is executed to produce the vectors:
  * transformers `LlamaForCausalLM.generate(...)` exactly as src/models/mllm/seed_x.py:121-136 calls it (the
    reference's src/models/mllm/modeling_llama_xformer.py is a copy of that class with the attention core swapped
    for xformers, which is not installed here), eager attention, fp32, seeded tiny weights;
  * the reference's own `AutoImageTokenGenerationProcessor` (src/models/mllm/generation.py) with a stub tokenizer
    that maps the image-token strings to fixed ids;
  * the reference's own `QwenResampler` (src/models/qwen_resampler.py) for the input and output resamplers.
Weights are NOT stored: tests rebuild them with `tiny_weights()` below from the same seeds.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

TINY = dict(vocab_size=640, hidden_size=256, intermediate_size=704, num_hidden_layers=2, num_attention_heads=2,
            rms_norm_eps=1e-6, rope_theta=10000.0)
N_IMG = 16                                    # image tokens per <img> block (= queries of the output resampler)
BOI, EOI = 600, 600 + N_IMG + 1
IMG_IDS = [BOI] + [601 + i for i in range(N_IMG)] + [EOI]
RES_IN = dict(grid_size=4, embed_dim=256, num_heads=4, kv_dim=128)     # sampler tokens (128) -> LLM width (256)
RES_OUT = dict(grid_size=4, embed_dim=128, num_heads=4, kv_dim=256)    # LLM hidden (256) -> sampler tokens (128)
MAX_NEW = 28


def tiny_weights(seed=1234):
    """Seeded LLaMA state dict with transformers key names (fp32)."""
    g = torch.Generator().manual_seed(seed)
    H, I, V, L = TINY["hidden_size"], TINY["intermediate_size"], TINY["vocab_size"], TINY["num_hidden_layers"]
    R = lambda *s, std=0.05: torch.randn(*s, generator=g) * std
    sd = {"model.embed_tokens.weight": R(V, H, std=0.5), "lm_head.weight": R(V, H, std=0.08),
          "model.norm.weight": 1.0 + R(H, std=0.1)}
    for i in range(L):
        p = f"model.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            sd[p + f"self_attn.{n}.weight"] = R(H, H, std=0.06)
        sd[p + "mlp.gate_proj.weight"] = R(I, H, std=0.06)
        sd[p + "mlp.up_proj.weight"] = R(I, H, std=0.06)
        sd[p + "mlp.down_proj.weight"] = R(H, I, std=0.04)
        sd[p + "input_layernorm.weight"] = 1.0 + R(H, std=0.1)
        sd[p + "post_attention_layernorm.weight"] = 1.0 + R(H, std=0.1)
    return sd


def sincos_2d(embed_dim, grid):
    """2-D sin/cos table of qwen_resampler.py:37-86 (w index first, then h), restated with torch."""
    def one(dim, pos):
        omega = 1.0 / 10000 ** (torch.arange(dim // 2, dtype=torch.float32) / (dim / 2.0))
        out = pos.reshape(-1)[:, None] * omega[None]
        return torch.cat([out.sin(), out.cos()], 1)
    gh, gw = torch.meshgrid(torch.arange(grid, dtype=torch.float32), torch.arange(grid, dtype=torch.float32),
                            indexing="ij")
    return torch.cat([one(embed_dim // 2, gw), one(embed_dim // 2, gh)], 1)


def tiny_resampler(cfg, seed):
    g = torch.Generator().manual_seed(seed)
    E, KV, Q = cfg["embed_dim"], cfg["kv_dim"], cfg["grid_size"] ** 2
    R = lambda *s, std=0.05: torch.randn(*s, generator=g) * std
    return {"pos_embed": sincos_2d(E, cfg["grid_size"]), "query": R(Q, E, std=0.3), "kv_proj.weight": R(E, KV, std=0.1),
            "attn.in_proj_weight": R(3 * E, E, std=0.08), "attn.in_proj_bias": R(3 * E, std=0.05),
            "attn.out_proj.weight": R(E, E, std=0.08), "attn.out_proj.bias": R(E, std=0.05),
            "ln_q.weight": 1.0 + R(E, std=0.1), "ln_q.bias": R(E, std=0.05),
            "ln_kv.weight": 1.0 + R(E, std=0.1), "ln_kv.bias": R(E, std=0.05)}


def tiny_prompt(seed=7):
    """[bos, text.., <img>, 16 placeholders, </img>, text.., <img>]: the trailing <img> starts the forced chain."""
    g = torch.Generator().manual_seed(seed)
    t = lambda n: torch.randint(3, 590, (n,), generator=g).tolist()
    ids = [1] + t(9) + [BOI] + IMG_IDS[1:-1] + [EOI] + t(5) + [BOI]
    mask = [False] * len(ids)
    for i in range(11, 11 + N_IMG):
        mask[i] = True
    image_embeds = torch.randn(1, N_IMG, RES_IN["kv_dim"], generator=g)
    return torch.tensor(ids), torch.tensor(mask), image_embeds


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


class _StubTokenizer:
    """encode() of the image-token string -> the fixed ids (the real one is a LlamaTokenizer with added tokens)."""
    def encode(self, s, add_special_tokens=False):
        assert s.startswith("<img>") and s.endswith("</img>")
        return list(IMG_IDS)


def main():
    from transformers import LlamaConfig, LlamaForCausalLM, LogitsProcessorList
    gen_mod = _load(os.path.join(REF, "src/models/mllm/generation.py"), "ref_generation")
    qr_mod = _load(os.path.join(REF, "src/models/qwen_resampler.py"), "ref_qwen_resampler")

    cfg = LlamaConfig(**TINY, max_position_embeddings=256, attn_implementation="eager", tie_word_embeddings=False,
                      bos_token_id=1, eos_token_id=2, pad_token_id=0)
    llm = LlamaForCausalLM(cfg).float().eval()
    sd = tiny_weights()
    missing, unexpected = llm.load_state_dict(sd, strict=False)
    assert not unexpected and all("rotary" in m or "inv_freq" in m for m in missing), (missing, unexpected)

    res_in = qr_mod.QwenResampler(**RES_IN).float().eval()
    res_out = qr_mod.QwenResampler(**RES_OUT).float().eval()
    sd_in, sd_out = tiny_resampler(RES_IN, 11), tiny_resampler(RES_OUT, 12)
    res_in.load_state_dict(sd_in)
    res_out.load_state_dict(sd_out)
    assert torch.allclose(res_in.pos_embed, sd_in["pos_embed"], atol=1e-6)     # our table == the reference's numpy one

    input_ids, ids_cmp_mask, image_embeds = tiny_prompt()
    out = {}
    with torch.no_grad():
        emb = llm.get_input_embeddings()(input_ids[None])
        lm_in = res_in(image_embeds)
        emb[ids_cmp_mask[None]] = lm_in.reshape(-1, emb.shape[-1])
        out["input_resampler_out"] = lm_in[0].numpy()
        for tag, eos in (("a", 2), ("b", None)):
            if eos is None:                   # second run: stop on a token the first run emitted after the image block
                eos = int(out["a_ids"][N_IMG + 4])
            proc = LogitsProcessorList([gen_mod.AutoImageTokenGenerationProcessor(_StubTokenizer(), N_IMG)])
            o = llm.generate(input_ids=input_ids[None], inputs_embeds=emb, output_hidden_states=True,
                             return_dict_in_generate=True, logits_processor=proc, num_beams=1, do_sample=False,
                             max_new_tokens=MAX_NEW, eos_token_id=eos, pad_token_id=0)
            seq = o.sequences[0]
            new = seq[input_ids.shape[0]:] if seq.shape[0] > MAX_NEW else seq
            hs = torch.cat([h[-1] for h in o.hidden_states], 1)[0, input_ids.shape[0]:]
            out[f"{tag}_ids"] = new.numpy()
            out[f"{tag}_hidden"] = hs.numpy()
            out[f"{tag}_eos"] = np.int64(eos)
        feats = torch.from_numpy(out["a_hidden"][:N_IMG])[None]
        out["output_resampler_out"] = res_out(feats)[0].numpy()
    path = os.path.join(ROOT, "tests/golden/mllm_tiny.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: getattr(v, "shape", v) for k, v in out.items()})
    print("a_ids", out["a_ids"].tolist())
    print("b_ids", out["b_ids"].tolist())


if __name__ == "__main__":
    sys.exit(main())
