"""MLLM pre-pass on MI355X: the producer of `ip_image_embeds` for the sampler.

Mirrors, for batch 1 and greedy decoding (the only mode the reference uses: `do_sample=False`, `num_beams=1`):
  * `ContinuousLVLM.generate`           reference src/models/mllm/seed_x.py:90-171
  * `LlamaForCausalLM` + KV cache       reference src/models/mllm/modeling_llama_xformer.py:97-314, 428-610
  * `AutoImageTokenGenerationProcessor` reference src/models/mllm/generation.py:19-30 (folded into the pick kernel)
  * `QwenResampler`                     reference src/models/qwen_resampler.py:87-145
  * the hand-off into the sampler       reference scripts/demo/gradio.py:85-109  (`mllm_prepass`)

Execution model: every weight is read once per generated token, so decoding is HBM-bound; one token step is a static
list of `DS_OP_LLM_*` launches (csrc/llm.hip) whose step-varying scalars live in a device-side state block, captured
once into a hipGraph and replayed per token.  The host looks at the device only every `poll_every` tokens (one 32-byte
copy) to see whether EOS was produced.  The prompt is one pass per layer: projections through the MFMA GEMMs
(`ops.gemm`), attention through the decode kernel in 16-row chunks (`prompt_path="mfma"`, 22 ms for 96 tokens at 13B
dims); `prompt_path="chunks"` runs it through the token kernels 16 rows at a time instead (166 ms; kept for A/B).

Host-side packing (once): q|k|v and gate|up projections stacked; the two RMSNorm gains of each layer stay vectors and are
applied in the GEMV prologue at the reference's rounding points (normalise in fp32, round to fp16, times the fp16 gain -
folding them into the fp16 weights would round once instead of twice and can flip greedy near-ties); QwenResampler: the constant query projection and the position-embedding contribution to the keys are
precomputed (they depend on weights only).  There is no CPU/PyTorch execution path: without the HIP library every call
raises.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib, ops
from .engine import Plan, make_op

Tensor = torch.Tensor
CHUNK = 16  # prompt rows per pass (llm.hip: M <= 16)


@dataclass
class LlamaConfig:
    vocab_size: int = 32330            # LLaMA-2 vocabulary + the MLLM's added image/box tokens
    hidden_size: int = 5120            # LLaMA-2-13B dims (the SEED-X agent the reference loads, gradio.py:256-257)
    intermediate_size: int = 13824
    num_hidden_layers: int = 40
    num_attention_heads: int = 40
    num_key_value_heads: Optional[int] = None
    rms_norm_eps: float = 1e-5
    rope_theta: float = 10000.0

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def kv_heads(self) -> int:
        return self.num_key_value_heads or self.num_attention_heads

    @classmethod
    def from_hf(cls, c) -> "LlamaConfig":
        g = lambda k, d=None: getattr(c, k, d) if not isinstance(c, dict) else c.get(k, d)
        return cls(g("vocab_size"), g("hidden_size"), g("intermediate_size"), g("num_hidden_layers"),
                   g("num_attention_heads"), g("num_key_value_heads"), g("rms_norm_eps", 1e-6),
                   g("rope_theta", 10000.0) or 10000.0)


def llama_param_shapes(cfg: LlamaConfig) -> Dict[str, tuple]:
    H, I, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
    kv = cfg.kv_heads * cfg.head_dim
    s = {"model.embed_tokens.weight": (V, H), "model.norm.weight": (H,), "lm_head.weight": (V, H)}
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        s[p + "self_attn.q_proj.weight"] = (H, H)
        s[p + "self_attn.k_proj.weight"] = (kv, H)
        s[p + "self_attn.v_proj.weight"] = (kv, H)
        s[p + "self_attn.o_proj.weight"] = (H, H)
        s[p + "mlp.gate_proj.weight"] = (I, H)
        s[p + "mlp.up_proj.weight"] = (I, H)
        s[p + "mlp.down_proj.weight"] = (H, I)
        s[p + "input_layernorm.weight"] = (H,)
        s[p + "post_attention_layernorm.weight"] = (H,)
    return s


def random_llama_state_dict(cfg: LlamaConfig, device, seed: int = 0) -> Dict[str, Tensor]:
    """Seeded fp16 weights at the real shapes, generated on the device (no checkpoints exist offline)."""
    g = torch.Generator(device=device).manual_seed(seed)
    sd = {}
    for name, shape in llama_param_shapes(cfg).items():
        if len(shape) == 1:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g, device=device)
        else:
            std = 0.5 if "embed_tokens" in name else 1.0 / math.sqrt(shape[1])
            t = torch.randn(shape, generator=g, device=device, dtype=torch.float16).float() * std
        sd[name] = t.to(torch.float16)
    return sd


class LlamaDecodeEngine:
    """Device-resident LLaMA decoder + KV cache + the captured one-token launch plan."""

    def __init__(self, cfg: LlamaConfig, sd: Dict[str, Tensor], device, max_positions: int = 1024,
                 max_new_tokens: int = 512, use_graph: bool = True, poll_every: int = 8, prompt_path: str = "mfma"):
        _lib.load()
        if prompt_path not in ("mfma", "chunks"):
            raise ValueError("prompt_path: 'mfma' (GEMM projections) or 'chunks' (16-row passes of the token kernels)")
        self.cfg, self.dev = cfg, torch.device(device)
        self.T_max, self.cap = int(max_positions), int(max_new_tokens)
        self.use_graph, self.poll_every = use_graph, max(1, int(poll_every))
        H, I, V, L = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size, cfg.num_hidden_layers
        D, Hq, Hkv = cfg.head_dim, cfg.num_attention_heads, cfg.kv_heads
        if D not in (64, 128):
            raise ValueError(f"head_dim {D}: the decode attention kernel is built for 64 and 128")
        if H % 8 or I % 8:
            raise ValueError("hidden and intermediate sizes must be multiples of 8")
        dev = self.dev
        f16 = lambda t: t.detach().to(device=dev, dtype=torch.float16).contiguous()

        stack = lambda ws: torch.cat([t.detach().to(dev) for t in ws], 0).to(torch.float16).contiguous()

        self.embed = f16(sd["model.embed_tokens.weight"])
        self.lm_head = f16(sd["lm_head.weight"])
        self.norm_g = f16(sd["model.norm.weight"])
        # q|k|v and gate|up are stacked (one weight stream per projection group); the RMSNorm gains stay separate vectors
        # applied in the GEMV prologue with the reference's rounding points (normalise, round to fp16, times the fp16 gain)
        self.wqkv, self.wo, self.wgu, self.wdown, self.g_in, self.g_post = [], [], [], [], [], []
        for i in range(L):
            p = f"model.layers.{i}."
            self.wqkv.append(stack([sd[p + f"self_attn.{n}_proj.weight"] for n in "qkv"]))
            self.g_in.append(f16(sd[p + "input_layernorm.weight"]))
            self.wo.append(f16(sd[p + "self_attn.o_proj.weight"]))
            self.wgu.append(stack([sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"]]))
            self.g_post.append(f16(sd[p + "post_attention_layernorm.weight"]))
            self.wdown.append(f16(sd[p + "mlp.down_proj.weight"]))
        E = lambda *s, dtype=torch.float16: torch.zeros(s, dtype=dtype, device=dev)
        self.qkv_dim = (Hq + 2 * Hkv) * D
        self.h, self.qkv, self.att = E(CHUNK, H), E(CHUNK, self.qkv_dim), E(CHUNK, Hq * D)
        self.act, self.hn, self.logits = E(CHUNK, I), E(1, H), E(V)
        self.kc = [E(self.T_max, Hkv * D) for _ in range(L)]
        self.vc = [E(self.T_max, Hkv * D) for _ in range(L)]
        inv_freq = 1.0 / (cfg.rope_theta ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
        fr = torch.outer(torch.arange(self.T_max, dtype=torch.float32), inv_freq)      # rotary table (weights-like)
        self.rope_cos, self.rope_sin = fr.cos().to(dev).contiguous(), fr.sin().to(dev).contiguous()
        self.state = E(8, dtype=torch.int32)
        self.state_pf = E(8, dtype=torch.int32)             # prompt pass: per-layer chunk cursor (see _prompt_mfma)
        self.prompt_path = prompt_path
        self.out_ids = E(self.cap, dtype=torch.int32)
        self.feat = E(self.cap, H)
        self.chain = E(1, dtype=torch.int32)
        self.n_chain = 0
        self._plans: Dict[tuple, Plan] = {}
        self._stream: Optional[torch.cuda.Stream] = None
        self.last_run_info: dict = {}

    @classmethod
    def from_pretrained_module(cls, llm, device, **kw) -> "LlamaDecodeEngine":
        """`llm`: a transformers/reference LlamaForCausalLM (anything with `.config` and `.state_dict()`)."""
        return cls(LlamaConfig.from_hf(llm.config), llm.state_dict(), device, **kw)

    def weight_bytes_per_token(self) -> int:
        """Algorithmic HBM bytes of one decode step: every layer matrix + lm_head once (+ one embedding row)."""
        n = sum(w.numel() for ws in (self.wqkv, self.wo, self.wgu, self.wdown) for w in ws) + self.lm_head.numel()
        return 2 * (n + self.cfg.hidden_size)

    def tensors(self) -> List[Tensor]:
        """Frozen weights in kernel layout (the multi-GPU weight broadcast list)."""
        return [self.embed, self.lm_head, self.norm_g] + self.wqkv + self.wo + self.wgu + self.wdown + self.g_in + self.g_post

    # ---- launch lists ------------------------------------------------------------------------------------
    def _ops(self, M: int, kind: str) -> list:
        """kind: 'chunk' (prompt rows, more follow), 'last' (final prompt chunk -> first token), 'token' (1 row)."""
        c = self.cfg
        H, I, V = c.hidden_size, c.intermediate_size, c.vocab_size
        D, Hq, Hkv = c.head_dim, c.num_attention_heads, c.kv_heads
        eps, scale = c.rms_norm_eps, 1.0 / math.sqrt(D)
        ops_ = []
        if kind == "token":
            ops_.append(make_op("LLM_EMBED", i=(H, V), p=(self.embed, self.state, self.h)))
        for l in range(c.num_hidden_layers):
            ops_.append(make_op("LLM_GEMV", i=(M, self.qkv_dim, H, 1, 0), f=(eps,), l=(H, self.qkv_dim, 0),
                                p=(self.h, self.wqkv[l], self.qkv, None, self.g_in[l])))
            ops_.append(make_op("LLM_ATTN", i=(M, Hq, Hkv, D, self.T_max), f=(scale,), l=(self.qkv_dim, Hkv * D, Hq * D),
                                p=(self.qkv, self.kc[l], self.vc[l], self.rope_cos, self.rope_sin, self.att,
                                   self.state)))
            ops_.append(make_op("LLM_GEMV", i=(M, H, Hq * D, 0, 0), f=(eps,), l=(Hq * D, H, H),
                                p=(self.att, self.wo[l], self.h, self.h)))
            ops_.append(make_op("LLM_GEMV", i=(M, I, H, 1, 1), f=(eps,), l=(H, I, 0),
                                p=(self.h, self.wgu[l], self.act, None, self.g_post[l])))
            ops_.append(make_op("LLM_GEMV", i=(M, H, I, 0, 0), f=(eps,), l=(I, H, H),
                                p=(self.act, self.wdown[l], self.h, self.h)))
        if kind == "chunk":
            ops_.append(make_op("LLM_ADVANCE", i=(M,), p=(self.state,)))
            return ops_
        last_row = self.h.data_ptr() + (M - 1) * H * 2
        ops_.append(make_op("LLM_RMSNORM", i=(1, H, self.cap), f=(eps,), l=(H, H),
                            p=(last_row, self.norm_g, self.hn, self.feat if kind == "token" else None, self.state)))
        ops_.append(make_op("LLM_GEMV", i=(1, V, H, 0, 0), f=(eps,), l=(H, V, 0),
                            p=(self.hn, self.lm_head, self.logits, None)))
        ops_.append(make_op("LLM_SELECT", i=(V, self.n_chain, self.cap, M),
                            p=(self.logits, self.chain if self.n_chain else None, self.state, self.out_ids)))
        return ops_

    def weights_changed(self) -> None:
        """The weight tensors moved or changed (multi-GPU broadcast into the weight arena): cached launch plans hold raw
        pointers and are rebuilt on next use."""
        self._plans.clear()

    def _plan(self, M: int, kind: str) -> Plan:
        key = (M, kind, self.n_chain, self.chain.data_ptr())
        pl = self._plans.get(key)
        if pl is None:
            pl = Plan(self._ops(M, kind), keep=[self])
            self._plans[key] = pl
        return pl

    def set_image_token_chain(self, img_ids_list: Optional[Sequence[int]]) -> None:
        """[<img>, <img_00000> .. <img_{n-1}>, </img>] of the logits processor (None/empty = plain greedy)."""
        ids = list(img_ids_list or [])
        if ids == getattr(self, "_chain_ids", None):
            return
        self._chain_ids = ids
        self.n_chain = len(ids)
        self.chain = torch.tensor(ids or [0], dtype=torch.int32, device=self.dev)
        self._plans.clear()

    # ---- generation --------------------------------------------------------------------------------------
    @torch.no_grad()
    def generate(self, inputs_embeds: Tensor, last_prompt_id: int, eos_token_id: int, max_new_tokens: int) -> dict:
        """Greedy decoding from prompt embeddings [T0, hidden] (fp16, device).  Returns the new ids [n] (int64, device)
        and `hidden` [n-1, hidden]: the post-final-norm state of every generated token that was fed back."""
        T0 = int(inputs_embeds.shape[0])
        H = self.cfg.hidden_size
        if inputs_embeds.shape[1] != H or inputs_embeds.dtype != torch.float16 or not inputs_embeds.is_cuda:
            raise ValueError("inputs_embeds must be a fp16 device tensor [T, hidden]")
        max_new = int(max_new_tokens)
        if not 0 < max_new <= self.cap:
            raise ValueError(f"max_new_tokens {max_new} outside (0, {self.cap}] (engine capacity)")
        if T0 < 1 or T0 + max_new > self.T_max:
            raise ValueError(f"prompt {T0} + max_new_tokens {max_new} exceeds the KV cache ({self.T_max} positions)")
        dev = self.dev
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=dev)
        st = self._stream
        st.wait_stream(torch.cuda.current_stream(dev))
        graph = False
        steps = 0
        with torch.cuda.stream(st):
            self.state.copy_(torch.tensor([0, 0, 0, int(last_prompt_id), max_new, int(eos_token_id), 0, 0],
                                          dtype=torch.int32), non_blocking=False)
            if T0 > CHUNK and self.prompt_path == "mfma":
                self._prompt_mfma(inputs_embeds)
            else:
                for r0 in range(0, T0, CHUNK):
                    m = min(CHUNK, T0 - r0)
                    self.h[:m].copy_(inputs_embeds[r0:r0 + m])
                    self._plan(m, "last" if r0 + m == T0 else "chunk").run(st.cuda_stream)
            tok = self._plan(1, "token")
            done = False
            while not done and steps < max_new - 1:
                burst = min(self.poll_every, max_new - 1 - steps)
                for _ in range(burst):
                    if self.use_graph and not tok.captured:
                        tok.run(st.cuda_stream)       # first token eager, then capture the launch list once
                        tok.capture(st.cuda_stream)
                    elif self.use_graph:
                        tok.replay(st.cuda_stream)
                        graph = True
                    else:
                        tok.run(st.cuda_stream)
                steps += burst
                done = bool(self.state[2].item())     # the only host<->device sync of the loop (every `poll_every`)
            n = int(self.state[1].item())
            ids = self.out_ids[:n].to(torch.int64)
            hidden = self.feat[:max(n - 1, 0)].clone()
        torch.cuda.current_stream(dev).wait_stream(st)
        self.last_run_info = {"graph": graph, "prompt_tokens": T0, "new_tokens": n, "token_steps_launched": steps,
                              "ops_per_token": tok.n}
        return {"ids": ids, "hidden": hidden}

    def _prompt_mfma(self, inputs_embeds: Tensor) -> None:
        """Whole prompt in one pass per layer: the projections are [T0,K] x [N,K]^T MFMA GEMMs (weights streamed once
        per layer instead of once per 16-row chunk); only the attention walks the rows in chunks of 16 (causal inside
        a chunk, cache rows before it).  Ends like the chunked path: final norm of the last row, lm_head, first pick."""
        c = self.cfg
        T0 = int(inputs_embeds.shape[0])
        Hq, Hkv, eps = c.num_attention_heads, c.kv_heads, c.rms_norm_eps
        scale = 1.0 / math.sqrt(c.head_dim)
        pf = self.state_pf
        h = inputs_embeds.contiguous().clone()
        att = torch.empty((T0, Hq * c.head_dim), dtype=torch.float16, device=self.dev)
        for l in range(c.num_hidden_layers):
            xn = ops.llm_rmsnorm(h, self.g_in[l], eps)                       # LlamaRMSNorm incl. its gain, then the plain GEMM
            qkv = ops.gemm(xn, self.wqkv[l])
            pf.zero_()                                                       # chunk cursor of this layer's cache
            for r0 in range(0, T0, CHUNK):
                m = min(CHUNK, T0 - r0)
                ops.llm_attention(qkv[r0:r0 + m], self.kc[l], self.vc[l], self.rope_cos, self.rope_sin, pf, Hq, Hkv,
                                  scale, out=att[r0:r0 + m])
                ops.llm_advance(pf, m)
            h = ops.gemm(att, self.wo[l], residual=h)
            xn = ops.llm_rmsnorm(h, self.g_post[l], eps)
            act = ops.llm_swiglu(ops.gemm(xn, self.wgu[l]))
            h = ops.gemm(act, self.wdown[l], residual=h)
        ops.llm_rmsnorm(h[T0 - 1:T0], self.norm_g, eps, out=self.hn)
        ops.llm_gemv(self.hn, self.lm_head, out=self.logits.view(1, -1))
        ops.llm_select(self.logits, self.chain if self.n_chain else None, T0, self.state, self.out_ids)

    def embed_tokens(self, input_ids: Tensor) -> Tensor:
        """Row gather from the embedding table (data movement only)."""
        return self.embed.index_select(0, input_ids.to(self.dev).view(-1).long())


def sincos_pos_embed_2d(embed_dim: int, grid_size: int) -> Tensor:
    """The fixed 2-D sin/cos table `QwenResampler.pos_embed` is initialised with (qwen_resampler.py:37-86): first half
    of the channels encodes the column index, second half the row index, each as [sin | cos] over 10000^(-2i/d)."""
    def one(dim, pos):
        omega = 1.0 / 10000 ** (torch.arange(dim // 2, dtype=torch.float32) / (dim / 2.0))
        out = pos.reshape(-1)[:, None] * omega[None]
        return torch.cat([out.sin(), out.cos()], 1)
    r = torch.arange(grid_size, dtype=torch.float32)
    gh, gw = torch.meshgrid(r, r, indexing="ij")
    return torch.cat([one(embed_dim // 2, gw), one(embed_dim // 2, gh)], 1)


def random_qwen_resampler_state_dict(grid_size: int, embed_dim: int, kv_dim: int, device, seed: int = 0) -> Dict[str, Tensor]:
    """Seeded weights with the reference module's key names (benchmarks / tests; no checkpoints exist offline)."""
    g = torch.Generator(device=device).manual_seed(seed)
    E, Q = embed_dim, grid_size ** 2
    R = lambda *s, std=0.02: torch.randn(*s, generator=g, device=device) * std
    return {"pos_embed": sincos_pos_embed_2d(E, grid_size).to(device), "query": R(Q, E),
            "kv_proj.weight": R(E, kv_dim, std=1.0 / math.sqrt(kv_dim)),
            "attn.in_proj_weight": R(3 * E, E, std=1.0 / math.sqrt(E)), "attn.in_proj_bias": R(3 * E),
            "attn.out_proj.weight": R(E, E, std=1.0 / math.sqrt(E)), "attn.out_proj.bias": R(E),
            "ln_q.weight": 1.0 + R(E), "ln_q.bias": R(E), "ln_kv.weight": 1.0 + R(E), "ln_kv.bias": R(E)}


class QwenResampler:
    """Single cross-attention resampler (reference src/models/qwen_resampler.py:87-145) on the HIP ops.
    Weight-only terms are folded at construction: q = (ln_q(query)+pos) Wq^T + bq is a constant, and the position
    embedding enters the keys as a per-token additive term pos Wk^T + bk."""

    def __init__(self, sd: Dict[str, Tensor], num_heads: int, device):
        _lib.load()
        dev = torch.device(device)
        f32 = lambda k: sd[k].detach().to(dev).float()
        self.dev, self.heads = dev, int(num_heads)
        query, pos = f32("query"), f32("pos_embed")
        self.num_queries, E = query.shape
        self.embed_dim = E
        wi, bi = f32("attn.in_proj_weight"), f32("attn.in_proj_bias")
        q = torch.nn.functional.layer_norm(query, (E,), f32("ln_q.weight"), f32("ln_q.bias")) + pos
        half = lambda t: t.to(torch.float16).contiguous()
        self.q = half(q @ wi[:E].T + bi[:E])[None]                                   # [1,Q,E]
        self.kv_proj = half(f32("kv_proj.weight")) if "kv_proj.weight" in sd else None
        self.ln_g, self.ln_b = half(f32("ln_kv.weight")), half(f32("ln_kv.bias"))
        self.w_kv = half(wi[E:])                                                     # [2E,E]: k rows then v rows
        self.pos = pos
        self._wk, self._bk, self._bv = wi[E:2 * E], bi[E:2 * E], bi[2 * E:]
        self._add: Dict[int, Tensor] = {}
        self.w_out, self.b_out = half(f32("attn.out_proj.weight")), half(f32("attn.out_proj.bias"))

    def weights_changed(self) -> None:
        """Drop the per-length key addends derived from the (rewritten) weights."""
        self._add.clear()

    def tensors(self) -> List[Tensor]:
        """Frozen (folded) weights (the multi-GPU weight broadcast list); the per-length key addends are derived."""
        self._add.clear()
        return [t for t in (self.q, self.kv_proj, self.ln_g, self.ln_b, self.w_kv, self.pos, self._wk, self._bk, self._bv,
                            self.w_out, self.b_out) if t is not None]

    def _kv_addend(self, L: int) -> Tensor:
        """[L,2E] = [pos Wk^T + bk | bv] (pos interpolated like get_abs_pos when L differs from the query grid)."""
        a = self._add.get(L)
        if a is None:
            pos = self.pos
            if L != pos.shape[0]:
                s, t = int(math.sqrt(pos.shape[0])), int(math.sqrt(L))
                pos = torch.nn.functional.interpolate(pos.reshape(1, s, s, -1).permute(0, 3, 1, 2), size=(t, t),
                                                      mode="bicubic", align_corners=False)
                pos = pos.permute(0, 2, 3, 1).flatten(0, 2)
            a = torch.cat([pos @ self._wk.T + self._bk, self._bv[None].expand(pos.shape[0], -1)], 1)
            a = a.to(torch.float16).contiguous()
            self._add[L] = a
        return a

    @torch.no_grad()
    def __call__(self, x: Tensor) -> Tensor:
        """x: [B, L, kv_dim] -> [B, num_queries, embed_dim] (fp16)."""
        B, L, _ = x.shape
        E = self.embed_dim
        x = x.to(device=self.dev, dtype=torch.float16).reshape(B * L, -1).contiguous()
        if self.kv_proj is not None:
            x = ops.gemm(x, self.kv_proj)
        x = ops.layernorm(x, self.ln_g, self.ln_b, 1e-5)
        add = self._kv_addend(L)
        kv = ops.gemm(x, self.w_kv, residual=add if B == 1 else add.repeat(B, 1)).view(B, L, 2 * E)
        q = self.q if B == 1 else self.q.expand(B, -1, -1).contiguous()
        o = ops.small_attention(q, kv[:, :, :E], kv[:, :, E:], self.heads, 1.0 / math.sqrt(E // self.heads))
        return ops.gemm(o.view(B * self.num_queries, E), self.w_out, bias=self.b_out).view(B, self.num_queries, E)


BOI_TOKEN, EOI_TOKEN, IMG_TOKEN = "<img>", "</img>", "<img_{:05d}>"


def image_token_ids(tokenizer, num_img_gen_tokens: int, img_ids_list: Optional[Sequence[int]] = None):
    """(processor chain, </img> id, the `num_img_gen_tokens` <img_xxxxx> ids) the way the reference derives them.

    The LLaMA sentencepiece tokenizer prepends a '▁' id to whatever it encodes; the reference therefore takes
    `encode(EOI_TOKEN)[1]` and `encode(img tokens)[1:]` (seed_x.py:139-141, gradio.py:44-45) but keeps the FULL encoded
    list, prefix included, as the logits processor's chain (generation.py:15-17).  Same here: the chain is the list as
    encoded; </img> is its last id and the image ids are the `num_img_gen_tokens` ids before it, so a list with or
    without the prefix gives the same answer."""
    if img_ids_list is None:
        if tokenizer is None:
            raise ValueError("pass a tokenizer or `img_ids_list`")
        s = BOI_TOKEN + "".join(IMG_TOKEN.format(i) for i in range(num_img_gen_tokens)) + EOI_TOKEN
        img_ids_list = tokenizer.encode(s, add_special_tokens=False)
    chain = [int(v) for v in img_ids_list]
    if len(chain) < num_img_gen_tokens + 2:
        raise ValueError(f"image-token chain has {len(chain)} ids; needs <img> + {num_img_gen_tokens} image ids + </img>")
    return chain, chain[-1], chain[-(num_img_gen_tokens + 1):-1]


class ContinuousLVLM:
    """`ContinuousLVLM.generate` of the reference (seed_x.py:90-171) over the decode engine.

    `tokenizer` is only used the way the reference uses it: to turn the image-token strings into ids and to decode the
    result; callers without tokenizer files pass `img_ids_list=[<img>, <img_00000>.., </img>]` instead."""

    def __init__(self, llm: LlamaDecodeEngine, input_resampler: QwenResampler, output_resampler: QwenResampler):
        self.llm, self.input_resampler, self.output_resampler = llm, input_resampler, output_resampler

    def dtype(self):
        return torch.float16

    def tensors(self) -> List[Tensor]:
        """Frozen weights of the whole agent: LLaMA decode engine + both QwenResamplers (multi-GPU broadcast list)."""
        return self.llm.tensors() + self.input_resampler.tensors() + self.output_resampler.tensors()

    def weights_changed(self) -> None:
        self.llm.weights_changed()
        self.input_resampler.weights_changed()
        self.output_resampler.weights_changed()

    @torch.no_grad()
    def generate(self, tokenizer=None, prompt=None, input_ids=None, image_embeds=None, ids_cmp_mask=None,
                 logits_processor=None, num_img_gen_tokens=64, temperature=0.7, num_beams=1, max_new_tokens=120,
                 top_p=0.5, img_ids_list: Optional[Sequence[int]] = None, eos_token_id: Optional[int] = None) -> dict:
        if logits_processor is not None:
            raise NotImplementedError("the image-token processor is built into the pick kernel; custom processors "
                                      "have no device implementation")
        if num_beams != 1:
            raise NotImplementedError("the reference decodes greedily (num_beams=1, do_sample=False)")
        img_ids_list, eoi_token_id, image_gen_id_list = image_token_ids(tokenizer, num_img_gen_tokens, img_ids_list)
        if eos_token_id is None:
            eos_token_id = getattr(tokenizer, "eos_token_id", None)
            if eos_token_id is None:
                raise ValueError("pass `eos_token_id` (or a tokenizer that has one)")
        if prompt is not None:
            input_ids = tokenizer(prompt, return_tensors="pt").input_ids
        if isinstance(input_ids, list):
            input_ids = torch.tensor(input_ids)
        ids = input_ids.view(-1)
        llm = self.llm
        emb = llm.embed_tokens(ids)
        if image_embeds is not None:
            assert ids_cmp_mask is not None
            lm = self.input_resampler(image_embeds)
            emb[ids_cmp_mask.view(-1).to(emb.device)] = lm.reshape(-1, emb.shape[-1])
        llm.set_image_token_chain(img_ids_list)
        g = llm.generate(emb, int(ids[-1]), int(eos_token_id), int(max_new_tokens))
        generate_ids, last_hidden_states = g["ids"].clone(), g["hidden"]
        image_gen_ids = torch.tensor(image_gen_id_list, dtype=generate_ids.dtype, device=generate_ids.device)
        eoi_indices = torch.where(generate_ids == eoi_token_id)[0].tolist()
        num_gen_imgs = len(eoi_indices)
        ids_gen_mask = torch.zeros_like(generate_ids, dtype=torch.bool)
        img_gen_feat = None
        if num_gen_imgs > 0:
            feats = []
            for e in eoi_indices:
                if e >= num_img_gen_tokens:
                    feats.append(last_hidden_states[e - num_img_gen_tokens:e])
                    generate_ids[e - num_img_gen_tokens:e] = image_gen_ids
                    ids_gen_mask[e - num_img_gen_tokens:e] = True
            img_gen_feat = self.output_resampler(torch.stack(feats)).contiguous()
        text = tokenizer.decode(generate_ids, skip_special_tokens=True) if tokenizer is not None else None
        return {"text": text, "output_ids": generate_ids, "img_gen_feat": img_gen_feat, "num_gen_imgs": num_gen_imgs,
                "ids_gen_mask": ids_gen_mask}


@torch.no_grad()
def mllm_prepass(pipeline, agent: ContinuousLVLM, input_ids: Tensor, ids_cmp_mask: Tensor, ip_images: list,
                 mllm_scale: float, tokenizer=None, img_ids_list: Optional[Sequence[int]] = None,
                 eos_token_id: Optional[int] = None, max_new_tokens: int = 500) -> Tensor:
    """reference scripts/demo/gradio.py:85-109: character tokens -> MLLM -> blended `ip_image_embeds`
    [max_num_ips, num_vision_tokens, dim] to pass to `pipeline(ip_images=[], ip_image_embeds=...)`."""
    cfg = pipeline.unet.config
    nv, n_ip = cfg.num_vision_tokens, cfg.max_num_ips
    image_embeds = pipeline.encode_ip_tokens(ip_images)[:, nv:, :]                   # [1, n_ip*nv, dim]
    out = agent.generate(tokenizer=tokenizer, input_ids=input_ids.unsqueeze(0), image_embeds=image_embeds,
                         ids_cmp_mask=ids_cmp_mask.unsqueeze(0), max_new_tokens=max_new_tokens,
                         num_img_gen_tokens=agent.output_resampler.num_queries, img_ids_list=img_ids_list,
                         eos_token_id=eos_token_id)
    if out["img_gen_feat"] is None:
        raise RuntimeError("the MLLM produced no image block")
    gen = out["img_gen_feat"].view(n_ip, nv, -1)
    base = image_embeds.reshape(n_ip, nv, -1).to(gen.dtype)
    return ops.blend(gen, base, float(mllm_scale))
