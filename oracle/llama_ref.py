"""TEST INFRASTRUCTURE ONLY — fp32 CPU restatement of the MLLM pre-pass.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(diffsensei_amd/mllm.py) never does.

Follows, function by function:
  * the LLaMA decoder the reference vendors in src/models/mllm/modeling_llama_xformer.py
      RMSNorm (transformers LlamaRMSNorm, imported at :95), rotary embedding :97-149, MLP :152-167,
      attention with KV cache :192-244 (causal over the prompt, unmasked single-query afterwards),
      decoder layer :247-314, model :428-610 (`all_hidden_states[-1]` is the hidden state AFTER the final norm,
      :595-599), lm_head :612-.
  * `AutoImageTokenGenerationProcessor.__call__`        src/models/mllm/generation.py:19-30
  * `ContinuousLVLM.generate` (greedy, `do_sample=False`) src/models/mllm/seed_x.py:90-171
  * `QwenResampler.forward`                             src/models/qwen_resampler.py:130-145
  * the hand-off into the sampler                        scripts/demo/gradio.py:85-109

Parity pin: tests/golden/mllm_tiny.npz was produced by oracle/make_golden_mllm.py, which runs transformers'
LlamaForCausalLM.generate (the class the reference's vendored file is a copy of) with the reference's own
logits processor, and the reference's own QwenResampler, on seeded tiny weights; tests/test_oracle_mllm.py holds
this restatement to those vectors.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


@dataclass
class LlamaRefConfig:
    vocab_size: int
    hidden_size: int
    intermediate_size: int
    num_hidden_layers: int
    num_attention_heads: int
    rms_norm_eps: float = 1e-6
    rope_theta: float = 10000.0

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads


def rms_norm(x: Tensor, w: Tensor, eps: float) -> Tensor:
    var = x.float().pow(2).mean(-1, keepdim=True)
    return w * (x.float() * torch.rsqrt(var + eps))


def rope_tables(head_dim: int, n_pos: int, theta: float = 10000.0) -> Tuple[Tensor, Tensor]:
    """cos/sin [n_pos, head_dim/2] (modeling_llama_xformer.py:101-114; both halves of `emb` are identical)."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2).float() / head_dim))
    freqs = torch.outer(torch.arange(n_pos).float(), inv_freq)
    return freqs.cos(), freqs.sin()


def apply_rope(x: Tensor, cos: Tensor, sin: Tensor) -> Tensor:
    """x: [T, heads, D]; cos/sin: [T, D/2].  q*cos + rotate_half(q)*sin with rotate_half = cat(-x2, x1)."""
    half = x.shape[-1] // 2
    x1, x2 = x[..., :half], x[..., half:]
    c, s = cos[:, None, :], sin[:, None, :]
    return torch.cat([x1 * c - x2 * s, x2 * c + x1 * s], dim=-1)


class KVCache:
    def __init__(self, cfg: LlamaRefConfig):
        self.k: List[Optional[Tensor]] = [None] * cfg.num_hidden_layers
        self.v: List[Optional[Tensor]] = [None] * cfg.num_hidden_layers
        self.length = 0


def llama_forward(sd: Dict[str, Tensor], cfg: LlamaRefConfig, x: Tensor, cache: KVCache) -> Tuple[Tensor, Tensor]:
    """Append the rows of x [T, hidden] to the cache; returns (post-final-norm hidden [T, hidden], logits [T, vocab])."""
    T = x.shape[0]
    Hh, D = cfg.num_attention_heads, cfg.head_dim
    pos0 = cache.length
    cos, sin = rope_tables(D, pos0 + T, cfg.rope_theta)
    cos, sin = cos[pos0:], sin[pos0:]
    h = x.float()
    for li in range(cfg.num_hidden_layers):
        p = f"model.layers.{li}."
        xn = rms_norm(h, sd[p + "input_layernorm.weight"].float(), cfg.rms_norm_eps)
        q = (xn @ sd[p + "self_attn.q_proj.weight"].float().T).view(T, Hh, D)
        k = (xn @ sd[p + "self_attn.k_proj.weight"].float().T).view(T, Hh, D)
        v = (xn @ sd[p + "self_attn.v_proj.weight"].float().T).view(T, Hh, D)
        q, k = apply_rope(q, cos, sin), apply_rope(k, cos, sin)
        if cache.k[li] is not None:
            k = torch.cat([cache.k[li], k], 0)
            v = torch.cat([cache.v[li], v], 0)
        cache.k[li], cache.v[li] = k, v
        s = torch.einsum("thd,shd->hts", q, k) / math.sqrt(D)
        # row t of this chunk sits at absolute position pos0+t and sees keys 0..pos0+t
        keep = torch.arange(pos0 + T)[None, :] <= (pos0 + torch.arange(T))[:, None]
        s = s.masked_fill(~keep[None], float("-inf"))
        a = torch.einsum("hts,shd->thd", s.softmax(-1), v).reshape(T, Hh * D)
        h = h + a @ sd[p + "self_attn.o_proj.weight"].float().T
        xn = rms_norm(h, sd[p + "post_attention_layernorm.weight"].float(), cfg.rms_norm_eps)
        g = xn @ sd[p + "mlp.gate_proj.weight"].float().T
        u = xn @ sd[p + "mlp.up_proj.weight"].float().T
        h = h + (F.silu(g) * u) @ sd[p + "mlp.down_proj.weight"].float().T
    cache.length = pos0 + T
    hn = rms_norm(h, sd["model.norm.weight"].float(), cfg.rms_norm_eps)
    return hn, hn @ sd["lm_head.weight"].float().T


def image_token_processor(prev_id: int, scores: Tensor, img_ids_list: Sequence[int]) -> Tensor:
    """generation.py:19-30 for one sequence.  img_ids_list = [<img>, <img_00000> ... <img_{n-1}>, </img>]."""
    scores = scores.clone()
    chain = list(img_ids_list)
    if prev_id in chain[:-1]:
        nxt = chain[chain.index(prev_id) + 1]
        scores[nxt] = scores.max() + 10.0
    else:
        scores[torch.tensor(chain[1:], dtype=torch.long)] = 0.0
    return scores


def greedy_generate(sd, cfg: LlamaRefConfig, inputs_embeds: Tensor, last_prompt_id: int, img_ids_list: Sequence[int],
                    eos_token_id: int, max_new_tokens: int) -> Dict[str, Tensor]:
    """transformers greedy search as driven by seed_x.py:121-136: the prompt goes in as `inputs_embeds`, every new token
    as its embedding row; stops after EOS or `max_new_tokens`.  Returns the new ids, the post-norm hidden state of every
    new token that was fed back (seed_x.py:143 keeps exactly those rows) and the top-2 margin of every choice."""
    cache = KVCache(cfg)
    emb = sd["model.embed_tokens.weight"].float()
    hn, logits = llama_forward(sd, cfg, inputs_embeds.float(), cache)
    ids: List[int] = []
    fed_hidden: List[Tensor] = []
    margins: List[float] = []
    prev = int(last_prompt_id)
    row = logits[-1]
    while True:
        sc = image_token_processor(prev, row, img_ids_list)
        top2 = sc.topk(2).values
        margins.append(float(top2[0] - top2[1]))
        nxt = int(sc.argmax())
        ids.append(nxt)
        if nxt == eos_token_id or len(ids) >= max_new_tokens:
            break
        hn, logits = llama_forward(sd, cfg, emb[nxt][None], cache)
        fed_hidden.append(hn[0])
        row = logits[0]
        prev = nxt
    H = cfg.hidden_size
    return {"ids": torch.tensor(ids, dtype=torch.long),
            "hidden": torch.stack(fed_hidden) if fed_hidden else torch.zeros(0, H),
            "margins": torch.tensor(margins)}


def qwen_resampler(sd: Dict[str, Tensor], x: Tensor, num_heads: int) -> Tensor:
    """qwen_resampler.py:130-145.  x: [B, L, kv_dim] with L == number of queries (no pos-embed interpolation);
    returns [B, Q, embed_dim].  nn.MultiheadAttention semantics: packed in_proj, scaled dot product, out_proj."""
    pos = sd["pos_embed"].float()
    Q, E = sd["query"].shape
    assert x.shape[1] == pos.shape[0], "restated for L == grid_size**2 only (get_abs_pos is then the identity)"
    if "kv_proj.weight" in sd:
        x = x.float() @ sd["kv_proj.weight"].float().T
    x = F.layer_norm(x.float(), (E,), sd["ln_kv.weight"].float(), sd["ln_kv.bias"].float())
    q = F.layer_norm(sd["query"].float(), (E,), sd["ln_q.weight"].float(), sd["ln_q.bias"].float())
    wi, bi = sd["attn.in_proj_weight"].float(), sd["attn.in_proj_bias"].float()
    qq = (q + pos) @ wi[:E].T + bi[:E]                     # [Q,E], shared by the batch
    kk = (x + pos[None]) @ wi[E:2 * E].T + bi[E:2 * E]     # [B,L,E]
    vv = x @ wi[2 * E:].T + bi[2 * E:]
    B, L, _ = x.shape
    d = E // num_heads
    qh = qq.view(Q, num_heads, d)
    kh, vh = kk.view(B, L, num_heads, d), vv.view(B, L, num_heads, d)
    s = torch.einsum("qhd,blhd->bhql", qh, kh) / math.sqrt(d)
    o = torch.einsum("bhql,blhd->bqhd", s.softmax(-1), vh).reshape(B, Q, E)
    return o @ sd["attn.out_proj.weight"].float().T + sd["attn.out_proj.bias"].float()


def lvlm_generate(llm_sd, cfg: LlamaRefConfig, in_res_sd, out_res_sd, res_heads: Tuple[int, int], input_ids: Tensor,
                  image_embeds: Optional[Tensor], ids_cmp_mask: Optional[Tensor], img_ids_list: Sequence[int],
                  eos_token_id: int, max_new_tokens: int, num_img_gen_tokens: int) -> Dict[str, object]:
    """seed_x.py:90-171 on token ids (the tokenizer only supplies `img_ids_list`, `</img>` and the decoded text)."""
    ids = input_ids.view(-1)
    emb = llm_sd["model.embed_tokens.weight"].float()[ids]
    if image_embeds is not None:
        lm = qwen_resampler(in_res_sd, image_embeds.float(), res_heads[0])
        emb = emb.clone()
        emb[ids_cmp_mask.view(-1)] = lm.reshape(-1, emb.shape[-1])
    g = greedy_generate(llm_sd, cfg, emb, int(ids[-1]), img_ids_list, eos_token_id, max_new_tokens)
    gen = g["ids"].clone()
    eoi = int(img_ids_list[-1])
    image_gen_ids = torch.tensor(list(img_ids_list[1:-1]), dtype=torch.long)
    eoi_idx = torch.where(gen == eoi)[0].tolist()
    ids_gen_mask = torch.zeros_like(gen, dtype=torch.bool)
    feats = []
    for e in eoi_idx:
        if e >= num_img_gen_tokens:
            feats.append(g["hidden"][e - num_img_gen_tokens:e])
            gen[e - num_img_gen_tokens:e] = image_gen_ids
            ids_gen_mask[e - num_img_gen_tokens:e] = True
    img_gen_feat = qwen_resampler(out_res_sd, torch.stack(feats), res_heads[1]) if feats else None
    return {"output_ids": gen, "img_gen_feat": img_gen_feat, "num_gen_imgs": len(eoi_idx),
            "ids_gen_mask": ids_gen_mask, "margins": g["margins"], "hidden": g["hidden"]}


def blend_ip_embeds(img_gen_feat: Tensor, image_embeds: Tensor, mllm_scale: float, max_num_ips: int,
                    num_vision_tokens: int) -> Tensor:
    """scripts/demo/gradio.py:108-109 -> the `ip_image_embeds` argument of the sampler."""
    a = img_gen_feat.view(max_num_ips, num_vision_tokens, -1)
    b = image_embeds.view(max_num_ips, num_vision_tokens, -1)
    return a * mllm_scale + b * (1 - mllm_scale)
