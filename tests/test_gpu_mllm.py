"""GPU: the MLLM pre-pass (csrc/llm.hip + diffsensei_amd/mllm.py) through the C ABI.

Per-kernel checks compare with plain PyTorch fp32 references of the same op computed from the same fp16-rounded
inputs; what is left is the kernel's fp16 output rounding (2^-11 relative) and accumulation order -> max |err| <=
4e-3 * max|ref| per op.  Whole-engine checks compare with the fp32 CPU oracle (oracle/llama_ref.py) and with the golden
vectors made by transformers + the reference's own modules (tests/golden/mllm_tiny.npz): token ids must be identical
wherever the oracle's top-2 margin exceeds the fp16 logit noise (5e-2), hidden states agree to 3e-2 (two fp16 layers).
"""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
GOLD = os.path.join(os.path.dirname(__file__), "golden", "mllm_tiny.npz")


def _h(shape, g, scale=1.0):
    return (torch.randn(shape, generator=g) * scale).half()


def _close(got, ref, tol=4e-3, what=""):
    got, ref = got.float().cpu(), ref.float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), what
    err = (got - ref).abs().max().item()
    den = max(ref.abs().max().item(), 1e-3)
    assert err <= tol * den, f"{what}: max err {err:.4g} vs max|ref| {den:.4g}"


# ------------------------------------------------------------------------------------------------ kernels
@pytest.mark.parametrize("M,N,K", [(1, 768, 256), (3, 520, 704), (16, 260, 5120), (21, 512, 1024), (1, 4096, 13824)])
def test_llm_gemv_plain_residual_rms(hip_lib, M, N, K):
    from diffsensei_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N + K)
    x, w, res = _h((M, K), g), _h((N, K), g, 1 / math.sqrt(K)), _h((M, N), g)
    xd, wd = x.to(DEV), w.to(DEV)
    _close(ops.llm_gemv(xd, wd), x.float() @ w.float().T, what="plain")
    y = res.to(DEV).clone()
    ops.llm_gemv(xd, wd, out=y, residual=y)                                   # in place: h += x W^T
    _close(y, (x.float() @ w.float().T).half().float() + res.float(), what="residual in place")
    r = torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + 1e-5)
    _close(ops.llm_gemv(xd, wd, rms=True, eps=1e-5), (x.float() @ w.float().T) * r, what="rms")
    # RMSNorm WITH its gain in the prologue, at the reference's rounding points (modeling_llama_xformer.py:77-82:
    # normalise in fp32, .to(fp16), weight * hidden in fp16) - a tight tolerance: the fp16 products are exact in fp32
    gain = (1.0 + 0.3 * torch.randn(K, generator=g)).half()
    xn = (gain.float() * (x.float() * r).half().float()).half()
    _close(ops.llm_gemv(xd, wd, rms=True, eps=1e-5, gain=gain.to(DEV)), xn.float() @ w.float().T, tol=1.5e-3,
           what="rms + gain")


@pytest.mark.parametrize("M,N,K", [(1, 704, 256), (5, 344, 512), (16, 1024, 1024)])
def test_llm_gemv_swiglu(hip_lib, M, N, K):
    from diffsensei_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    x, w = _h((M, K), g), _h((2 * N, K), g, 2 / math.sqrt(K))
    r = torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + 1e-6)
    gate, up = (x.float() @ w[:N].float().T) * r, (x.float() @ w[N:].float().T) * r
    _close(ops.llm_gemv(x.to(DEV), w.to(DEV), rms=True, swiglu=True, eps=1e-6), F.silu(gate) * up, tol=6e-3,
           what="swiglu")
    gain = (1.0 + 0.3 * torch.randn(K, generator=g)).half()
    xn = (gain.float() * (x.float() * r).half().float()).half().float()
    gate, up = (xn @ w[:N].float().T).half().float(), (xn @ w[N:].float().T).half().float()
    _close(ops.llm_gemv(x.to(DEV), w.to(DEV), rms=True, swiglu=True, eps=1e-6, gain=gain.to(DEV)),
           F.silu(gate).half().float() * up, tol=6e-3, what="swiglu + gain")


def _ref_attention(q, k, v, pos_q, theta=10000.0):
    """fp32: rotary (rotate_half form) + causal softmax attention; q/k/v [T, heads, D]; keys at positions 0..T-1."""
    from oracle import llama_ref as R
    T, Hh, D = k.shape
    cos, sin = R.rope_tables(D, T, theta)
    kr = R.apply_rope(k.float(), cos, sin).half().float()                       # the cache holds fp16 rotated keys
    qr = R.apply_rope(q.float(), cos[pos_q], sin[pos_q])
    s = torch.einsum("thd,shd->hts", qr, kr) / math.sqrt(D)
    keep = torch.arange(T)[None, :] <= pos_q[:, None]
    p = s.masked_fill(~keep[None], float("-inf")).softmax(-1).half().float()
    return torch.einsum("hts,shd->thd", p, v.float()).reshape(len(pos_q), Hh * D), kr


@pytest.mark.parametrize("D,heads,kv_heads", [(128, 3, 3), (64, 4, 2)])
def test_llm_attention_chunks_and_tokens(hip_lib, D, heads, kv_heads):
    """Prompt in chunks of 16 + 5 rows, then 3 single tokens, against one causal fp32 attention over all 24 rows."""
    from diffsensei_amd import ops
    g = torch.Generator().manual_seed(D + heads)
    T, T_max = 24, 40
    qkv = _h((T, (heads + 2 * kv_heads) * D), g)
    rep = heads // kv_heads
    q = qkv[:, :heads * D].view(T, heads, D)
    k = qkv[:, heads * D:(heads + kv_heads) * D].view(T, kv_heads, D)
    v = qkv[:, (heads + kv_heads) * D:].view(T, kv_heads, D)
    ref, kr = _ref_attention(q, k.repeat_interleave(rep, 1), v.repeat_interleave(rep, 1), torch.arange(T))
    inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2).float() / D))
    fr = torch.outer(torch.arange(T_max).float(), inv)
    cos, sin = fr.cos().to(DEV).contiguous(), fr.sin().to(DEV).contiguous()
    kc = torch.zeros(T_max, kv_heads * D, dtype=torch.float16, device=DEV)
    vc = torch.zeros_like(kc)
    state = torch.zeros(8, dtype=torch.int32, device=DEV)
    qd = qkv.to(DEV)
    outs, r0 = [], 0
    for m in (16, 5, 1, 1, 1):
        outs.append(ops.llm_attention(qd[r0:r0 + m].contiguous(), kc, vc, cos, sin, state, heads, kv_heads,
                                      1.0 / math.sqrt(D)))
        ops.llm_advance(state, m)
        r0 += m
    assert int(state[0]) == T
    _close(torch.cat(outs), ref, what="attention rows")
    _close(kc[:T].view(T, kv_heads, D), kr[:, ::rep], tol=2e-3, what="rotated key cache")
    assert torch.equal(vc[:T].cpu(), qkv[:, (heads + kv_heads) * D:]), "value cache must be a bit copy"
    assert not kc[T:].any() and not vc[T:].any(), "rows past the cache length were written"


def test_llm_rmsnorm_and_feature_tap(hip_lib):
    from diffsensei_amd import ops
    g = torch.Generator().manual_seed(3)
    x, gam = _h((5, 512), g, 3.0), (1 + 0.1 * torch.randn(512, generator=g)).half()
    ref = gam.float() * (x.float() * torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + 1e-6)).half().float()
    _close(ops.llm_rmsnorm(x.to(DEV), gam.to(DEV), 1e-6), ref, tol=2e-3, what="rmsnorm")
    feat = torch.zeros(4, 512, dtype=torch.float16, device=DEV)
    state = torch.tensor([0, 3, 0, 0, 9, 2, 0, 0], dtype=torch.int32, device=DEV)       # 3 ids out -> feature row 2
    y = ops.llm_rmsnorm(x[:1].to(DEV), gam.to(DEV), 1e-6, feat=feat, state=state)
    assert torch.equal(feat[2], y[0]) and not feat[[0, 1, 3]].any()
    state[2] = 1                                                                        # finished: the tap is closed
    feat.zero_()
    ops.llm_rmsnorm(x[:1].to(DEV), gam.to(DEV), 1e-6, feat=feat, state=state)
    assert not feat.any()


def test_llm_select_processor_semantics(hip_lib):
    """generation.py:19-30 on the device: forced chain, zeroing of image ids, lowest-id ties, EOS / max_new flags."""
    from diffsensei_amd import ops
    V = 3000
    chain = torch.tensor([2900, 2901, 2902, 2903], dtype=torch.int32, device=DEV)       # <img>, img_0, img_1, </img>
    out_ids = torch.zeros(8, dtype=torch.int32, device=DEV)

    def run(logits, prev, n_out=0, max_new=8, eos=2, use_chain=True):
        st = torch.tensor([10, n_out, 0, prev, max_new, eos, 0, 0], dtype=torch.int32, device=DEV)
        ops.llm_select(logits.half().to(DEV), chain if use_chain else None, 1, st, out_ids)
        return st.tolist()

    g = torch.Generator().manual_seed(0)
    base = -torch.rand(V, generator=g) - 0.5                                            # everything negative
    st = run(base, prev=5)
    assert st[3] == 2901 and out_ids[0] == 2901, "image ids (not <img>) are set to 0.0 -> lowest such id wins"
    assert st[:3] == [11, 1, 0]
    lg = base.clone(); lg[1234] = 0.0; lg[77] = 0.0
    assert run(lg, prev=5)[3] == 77, "ties go to the lowest id"
    lg = base.clone(); lg[2000] = 4.0
    assert run(lg, prev=5)[3] == 2000
    lg[2902] = 9.0                                                                      # would win, but it is zeroed
    assert run(lg, prev=5)[3] == 2000
    assert run(lg, prev=2900)[3] == 2901 and run(lg, prev=2901)[3] == 2902 and run(lg, prev=2902)[3] == 2903
    assert run(lg, prev=2903)[3] == 2000, "</img> is not a chain member"
    assert run(lg, prev=5, use_chain=False)[3] == 2902, "no processor: plain argmax"
    st = run(lg, prev=5, eos=2000, n_out=3)
    assert st[2] == 1 and st[1] == 4 and out_ids[3] == 2000, "EOS raises the finished flag after appending"
    assert run(lg, prev=5, n_out=7, max_new=8)[2] == 1 and run(lg, prev=5, n_out=6, max_new=8)[2] == 0
    st = torch.tensor([10, 4, 1, 5, 8, 2, 0, 0], dtype=torch.int32, device=DEV)
    before = out_ids.clone()
    ops.llm_select(lg.half().to(DEV), chain, 1, st, out_ids)
    assert st.tolist() == [10, 4, 1, 5, 8, 2, 0, 0] and torch.equal(before, out_ids), "finished -> no-op"


def test_llm_swiglu(hip_lib):
    from diffsensei_amd import ops
    g = torch.Generator().manual_seed(5)
    gu = _h((7, 2 * 704), g, 2.0)
    _close(ops.llm_swiglu(gu.to(DEV)), F.silu(gu[:, :704].float()) * gu[:, 704:].float(), tol=2e-3, what="swiglu")


def test_blend(hip_lib):
    from diffsensei_amd import ops
    g = torch.Generator().manual_seed(1)
    a, b = _h((4, 16, 64), g), _h((4, 16, 64), g)
    _close(ops.blend(a.to(DEV), b.to(DEV), 0.3), a.float() * 0.3 + b.float() * 0.7, tol=2e-3)


# ------------------------------------------------------------------------------------------------ engine
@pytest.fixture(scope="module")
def tiny(hip_lib):
    from oracle import make_golden_mllm as G
    from diffsensei_amd.mllm import ContinuousLVLM, LlamaConfig, LlamaDecodeEngine, QwenResampler
    cfg = LlamaConfig(vocab_size=G.TINY["vocab_size"], hidden_size=G.TINY["hidden_size"],
                      intermediate_size=G.TINY["intermediate_size"], num_hidden_layers=G.TINY["num_hidden_layers"],
                      num_attention_heads=G.TINY["num_attention_heads"], rms_norm_eps=G.TINY["rms_norm_eps"])
    sd = G.tiny_weights()
    sd_in, sd_out = G.tiny_resampler(G.RES_IN, 11), G.tiny_resampler(G.RES_OUT, 12)
    mk = lambda graph, path="mfma": LlamaDecodeEngine(cfg, sd, DEV, max_positions=96, max_new_tokens=40, use_graph=graph,
                                                      poll_every=4, prompt_path=path)
    res_in, res_out = QwenResampler(sd_in, G.RES_IN["num_heads"], DEV), QwenResampler(sd_out, G.RES_OUT["num_heads"], DEV)
    return {"G": G, "sd": sd, "sd_in": sd_in, "sd_out": sd_out, "mk": mk, "res_in": res_in, "res_out": res_out,
            "LVLM": ContinuousLVLM, "gold": dict(np.load(GOLD))}


def test_qwen_resampler_vs_reference_vectors(tiny):
    G, gold = tiny["G"], tiny["gold"]
    _, _, image_embeds = G.tiny_prompt()
    _close(tiny["res_in"](image_embeds.to(DEV))[0], torch.from_numpy(gold["input_resampler_out"]), tol=6e-3,
           what="input resampler")
    feats = torch.from_numpy(gold["a_hidden"][:G.N_IMG])[None]
    _close(tiny["res_out"](feats.to(DEV))[0], torch.from_numpy(gold["output_resampler_out"]), tol=6e-3,
           what="output resampler")
    x2 = torch.cat([image_embeds, image_embeds.flip(1)], 0)                                 # batch of 2
    y2 = tiny["res_in"](x2.to(DEV))
    _close(y2[0], torch.from_numpy(gold["input_resampler_out"]), tol=6e-3, what="batched row 0")


@pytest.mark.parametrize("E,KV", [(5120, 2048), (2048, 5120)])
def test_qwen_resampler_at_the_agent_dimensions(hip_lib, E, KV):
    """configs/model/diffsensei.yaml agent.{input,output}_resampler: 64 queries, 32 heads (head_dim 160 / 64)."""
    from diffsensei_amd.mllm import QwenResampler, random_qwen_resampler_state_dict
    from oracle import llama_ref as R
    sd = random_qwen_resampler_state_dict(8, E, KV, DEV, seed=E)
    x = _h((1, 64, KV), torch.Generator().manual_seed(KV))
    ref = R.qwen_resampler({k: v.float().cpu() for k, v in sd.items()}, x.float(), 32)
    _close(QwenResampler(sd, 32, DEV)(x.to(DEV)), ref, tol=1e-2, what=f"resampler {KV}->{E}")


def _check_ids(got, want, margins, what):
    """identical, except that a choice whose fp32 top-2 margin is inside the fp16 logit noise may legitimately flip
    (everything after such a flip is a different continuation)."""
    got, want = list(got), list(want)
    for i, (a, b) in enumerate(zip(got, want)):
        if a != b:
            assert margins[i] < 5e-2, f"{what}: id {i} is {a}, oracle {b} (margin {margins[i]:.3g})"
            return i
    assert len(got) == len(want), f"{what}: {len(got)} ids vs {len(want)}"
    return len(want)


@pytest.mark.parametrize("path", ["mfma", "chunks"])      # prompt pass: GEMM projections / 16-row passes of the token kernels
@pytest.mark.parametrize("graph", [True, False])
@pytest.mark.parametrize("tag", ["a", "b"])
def test_generate_matches_golden_and_oracle(tiny, graph, tag, path):
    from oracle import llama_ref as R
    G, gold = tiny["G"], tiny["gold"]
    input_ids, mask, image_embeds = G.tiny_prompt()
    eos = int(gold[f"{tag}_eos"])
    ref = R.lvlm_generate(tiny["sd"], R.LlamaRefConfig(**G.TINY), tiny["sd_in"], tiny["sd_out"],
                          (G.RES_IN["num_heads"], G.RES_OUT["num_heads"]), input_ids, image_embeds, mask, G.IMG_IDS, eos,
                          G.MAX_NEW, G.N_IMG)
    assert ref["output_ids"].tolist() == gold[f"{tag}_ids"].tolist()                      # oracle == transformers
    agent = tiny["LVLM"](tiny["mk"](graph, path), tiny["res_in"], tiny["res_out"])
    for rep in range(2):                                                                  # 2nd call reuses the plan/graph
        out = agent.generate(input_ids=input_ids[None], image_embeds=image_embeds.to(DEV), ids_cmp_mask=mask[None],
                             num_img_gen_tokens=G.N_IMG, max_new_tokens=G.MAX_NEW, img_ids_list=G.IMG_IDS,
                             eos_token_id=eos)
        n_same = _check_ids(out["output_ids"].tolist(), gold[f"{tag}_ids"].tolist(), ref["margins"].tolist(),
                            f"{tag}/graph={graph}/rep={rep}")
        assert n_same > G.N_IMG, "the forced image chain and </img> must always match"
        info = agent.llm.last_run_info
        assert info["prompt_tokens"] == len(input_ids) and info["graph"] == (graph and (rep > 0 or info["new_tokens"] > 2))
        hid = agent.llm.feat[:n_same - 1]
        _close(hid, torch.from_numpy(gold[f"{tag}_hidden"][:n_same - 1]), tol=3e-2, what="fed-back hidden states")
        assert out["num_gen_imgs"] == 1 and bool(out["ids_gen_mask"][:G.N_IMG].all())
        _close(out["img_gen_feat"][0], torch.from_numpy(gold["output_resampler_out"]), tol=3e-2, what="img_gen_feat")


def test_generate_with_sentencepiece_prefix_tokenizer(tiny):
    """`generate(tokenizer=...)` with a LLaMA-style tokenizer whose encode() prepends the '▁' id (the reference indexes
    past it: seed_x.py:139-141): same ids / features as passing the un-prefixed list, the prefix id stays in the chain."""
    G, gold = tiny["G"], tiny["gold"]
    input_ids, mask, image_embeds = G.tiny_prompt()
    eos = int(gold["a_eos"])
    prefix_id = 595                                              # an id the tiny vocabulary never generates here

    class Tok:
        eos_token_id = eos

        def encode(self, s, add_special_tokens=False):
            assert s.startswith("<img>") and s.endswith("</img>")
            return [prefix_id] + list(G.IMG_IDS)

        def decode(self, ids, skip_special_tokens=True):
            return " ".join(str(int(i)) for i in ids)

    agent = tiny["LVLM"](tiny["mk"](False), tiny["res_in"], tiny["res_out"])
    kw = dict(input_ids=input_ids[None], image_embeds=image_embeds.to(DEV), ids_cmp_mask=mask[None],
              num_img_gen_tokens=G.N_IMG, max_new_tokens=G.MAX_NEW)
    a = agent.generate(tokenizer=Tok(), **kw)
    assert agent.llm._chain_ids == [prefix_id] + list(G.IMG_IDS)
    b = agent.generate(img_ids_list=G.IMG_IDS, eos_token_id=eos, **kw)
    # the forced image block + </img> and everything derived from it are identical; the free continuation after it may
    # differ legitimately (with the prefix in the chain the processor also zeroes the <img> logit, generation.py:28)
    n = G.N_IMG + 1
    assert a["output_ids"][:n].tolist() == b["output_ids"][:n].tolist() == list(G.IMG_IDS[1:])
    assert a["num_gen_imgs"] >= 1 and bool(a["ids_gen_mask"][:G.N_IMG].all())
    assert torch.equal(a["img_gen_feat"][0], b["img_gen_feat"][0])
    assert a["text"] == " ".join(str(int(i)) for i in a["output_ids"])


def test_generate_argument_checks(tiny):
    eng = tiny["mk"](False)
    emb = torch.zeros(10, eng.cfg.hidden_size, dtype=torch.float16, device=DEV)
    with pytest.raises(ValueError):
        eng.generate(emb, 1, 2, 41)                      # above the engine's max_new_tokens capacity
    with pytest.raises(ValueError):
        eng.generate(torch.zeros(90, eng.cfg.hidden_size, dtype=torch.float16, device=DEV), 1, 2, 20)   # cache too small
    with pytest.raises(ValueError):
        eng.generate(emb.float(), 1, 2, 4)
