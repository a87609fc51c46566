"""Thin torch-tensor wrappers over the C ABI (one function per entry point of include/diffsensei_hip.h).

PyTorch is plumbing here: device memory and the current HIP stream.  No arithmetic happens in this file;
every function checks devices/dtypes, hands raw pointers to the library and raises on a non-zero status.
Layouts: activations fp16 channels-last — images [B, H*W, C] (or [B,H,W,C]), token matrices [rows, C].
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import check

Tensor = torch.Tensor


def bind_device(dev: torch.device) -> None:
    """One process drives one GPU.  The library launches on "the current stream", which HIP resolves per
    CURRENT device, so before a launch the tensors' device is made the current one (a no-op in the normal case where
    `init_from_env` / the caller already selected it).  NOTE: this changes process-global state (torch's current device) and
    does not restore it - deliberate under the one-process-one-GPU design; the library's per-device one-off set-up
    (`ds_first_on_device`, csrc/ds_common.h) is keyed by device, so a process that does touch a second GPU stays correct."""
    if dev.type == "cuda" and dev.index is not None and dev.index != torch.cuda.current_device():
        torch.cuda.set_device(dev)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _chk(*ts: Optional[Tensor], dtype=torch.float16) -> None:
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.DiffSenseiHipError("diffsensei_amd ops need CUDA(HIP) tensors — there is no CPU path")
        bind_device(t.device)
        if dtype is not None and t.dtype != dtype:
            raise _lib.DiffSenseiHipError(f"expected {dtype}, got {t.dtype}")
        if not t.is_contiguous():
            raise _lib.DiffSenseiHipError("tensor must be contiguous")


_EPILOGUES = {None: 0, "geglu": 1, "gelu": 2, "quick_gelu": 3}


def _geglu_code(geglu) -> int:
    """`geglu` argument of the GEMM wrappers -> epilogue code: False 0, True 1 (w / bias packed by `engine.pack_geglu`, 128-row
    groups), 320 -> 4 (packed by `engine.pack_geglu320`: gemm_g320_kernel)."""
    if geglu is True or geglu == 1:
        return 1
    if geglu == 320:
        return 4
    assert not geglu, geglu
    return 0


def gemm(x: Tensor, w: Tensor, bias: Optional[Tensor] = None, residual: Optional[Tensor] = None,
         geglu: bool = False, out: Optional[Tensor] = None, x2: Optional[Tensor] = None,
         act: Optional[str] = None) -> Tensor:
    """y = act(cat([x, x2], -1) @ w.T + bias) (+residual);  geglu: w/bias packed by `pack_geglu`, y = h*gelu(g)."""
    _chk(x, w, bias, residual, x2)
    epi = _geglu_code(geglu) if geglu else _EPILOGUES[act]
    M, K1 = x.shape
    K = K1 + (x2.shape[1] if x2 is not None else 0)
    N = w.shape[0]
    assert w.shape[1] == K
    n_out = N // 2 if geglu else N
    if out is None:
        out = torch.empty((M, n_out), dtype=torch.float16, device=x.device)
    L = _lib.load()
    check(L.ds_gemm_f16(_p(x), K1, _p(x2), 0 if x2 is None else x2.shape[1], K1, _p(w), K, _p(bias), _p(residual),
                        n_out, _p(out), n_out, M, N, K, epi, _stream()), "ds_gemm_f16")
    return out


def gemm_ln(x: Tensor, gw: Tensor, bias_ln: Optional[Tensor] = None, ln_c: Optional[Tensor] = None,
            ln_stats: Optional[Tensor] = None, residual: Optional[Tensor] = None, geglu: bool = False,
            emit_stats: bool = False, out: Optional[Tensor] = None):
    """The two roles of a fused LayerNorm (include/diffsensei_hip.h, ds_gemm_ln_f16):
    consumer (`ln_stats` [M,2] fp32 (mean, rstd), `ln_c` [N,2] f16, `gw`, `bias_ln` from `engine.pack_ln_fused`):
        y = rstd (x gw^T - mean c) + b' on the RAW x;
    producer (`emit_stats`): y = x gw^T + bias (+ residual) and the [N/64, M, 2] fp32 partial (sum, sum of squares) of the stored
        rows -> returns (y, partial)."""
    _chk(x, gw, bias_ln, ln_c, residual)
    _chk(ln_stats, dtype=torch.float32)
    M, K = x.shape
    N = gw.shape[0]
    n_out = N // 2 if geglu else N
    if out is None:
        out = torch.empty((M, n_out), dtype=torch.float16, device=x.device)
    part = torch.empty((N // 64, M, 2), dtype=torch.float32, device=x.device) if emit_stats else None
    L = _lib.load()
    check(L.ds_gemm_ln_f16(_p(x), K, _p(gw), K, _p(bias_ln), _p(ln_stats), _p(ln_c), _p(residual), n_out, _p(out), n_out,
                           _p(part), M, N, K, _geglu_code(geglu), _stream()), "ds_gemm_ln_f16")
    return (out, part) if emit_stats else out


def ln_finalize(partial: Tensor, C: int, eps: float = 1e-5) -> Tensor:
    """[strips, M, 2] fp32 partial sums of a producer GEMM -> [M, 2] fp32 (mean, rstd) over C columns."""
    _chk(partial, dtype=torch.float32)
    strips, M, _ = partial.shape
    st = torch.empty((M, 2), dtype=torch.float32, device=partial.device)
    check(_lib.load().ds_ln_finalize(_p(partial), _p(st), M, strips, C, eps, _stream()), "ds_ln_finalize")
    return st


def gemm_ln_fusable(M: int, N: int, K: int, geglu: bool = False, batch: int = 1) -> int:
    """Which fused-LayerNorm form the dispatch gives this GEMM: 1 = the 256 x 256 kernel (`gemm_ln` with finalised statistics),
    2 = the 128-wide kernels (`gemm_ln_partial` consumers, `gemm_ln` producers), 0 = none."""
    return int(_lib.load().ds_gemm_ln_fusable(M, N, K, _geglu_code(geglu), batch))


def gemm_ln_partial(x: Tensor, gw: Tensor, bias_ln: Optional[Tensor], ln_c: Tensor, partial: Tensor, eps: float = 1e-5,
                    residual: Optional[Tensor] = None, geglu: bool = False, emit_stats: bool = False,
                    out: Optional[Tensor] = None):
    """Consumer of a fused LayerNorm on the 128-wide kernels (ds_gemm_ln_partial_f16): `partial` [K/64, M, 2] fp32 is what a
    producer emitted; every block sums the partials of its own rows, no finalize launch.  y = rstd (x gw^T - mean c) + b'."""
    _chk(x, gw, bias_ln, ln_c, residual)
    _chk(partial, dtype=torch.float32)
    M, K = x.shape
    N = gw.shape[0]
    assert tuple(partial.shape) == (K // 64, M, 2), (tuple(partial.shape), (K // 64, M, 2))
    n_out = N // 2 if geglu else N
    if out is None:
        out = torch.empty((M, n_out), dtype=torch.float16, device=x.device)
    part = torch.empty((N // 64, M, 2), dtype=torch.float32, device=x.device) if emit_stats else None
    check(_lib.load().ds_gemm_ln_partial_f16(_p(x), K, _p(gw), K, _p(bias_ln), _p(partial), eps, _p(ln_c), _p(residual), n_out,
                                             _p(out), n_out, _p(part), M, N, K, _geglu_code(geglu), _stream()),
          "ds_gemm_ln_partial_f16")
    return (out, part) if emit_stats else out


def gemm_ln_swapped_partial(a: Tensor, x: Tensor, partial: Tensor, ln_cb: Tensor, eps: float = 1e-5,
                            out: Optional[Tensor] = None) -> Tensor:
    """`gemm_ln_swapped` on the 128-wide kernels (ds_gemm_ln_swapped_partial_f16): partial [K/64, Z*N, 2] fp32 as a producer
    emitted them for the rows of x.view(Z*N, K); no finalize launch."""
    _chk(a, x, ln_cb)
    _chk(partial, dtype=torch.float32)
    Z, N, K = x.shape
    M = a.shape[0]
    assert tuple(partial.shape) == (K // 64, Z * N, 2), tuple(partial.shape)
    if out is None:
        out = torch.empty((Z, M, N), dtype=torch.float16, device=x.device)
    check(_lib.load().ds_gemm_ln_swapped_partial_f16(_p(a), K, _p(x), K, N * K, _p(partial), eps, Z * N, N, _p(ln_cb), _p(out), N,
                                                     M * N, M, N, K, Z, _stream()), "ds_gemm_ln_swapped_partial_f16")
    return out


def gemm_ln_swapped(a: Tensor, x: Tensor, ln_stats: Tensor, ln_cb: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """out[z] = a @ LN(x[z])^T with the LayerNorm folded in (ds_gemm_ln_swapped_f16): a = gamma (.) W [M,K], x [Z,N,K] raw,
    ln_stats [Z*N,2] fp32 (mean, rstd), ln_cb [M,4] f16 (-c hi, -c lo, b' hi, b' lo) -> [Z,M,N]."""
    _chk(a, x, ln_cb)
    _chk(ln_stats, dtype=torch.float32)
    Z, N, K = x.shape
    M = a.shape[0]
    if out is None:
        out = torch.empty((Z, M, N), dtype=torch.float16, device=x.device)
    check(_lib.load().ds_gemm_ln_swapped_f16(_p(a), K, _p(x), K, N * K, _p(ln_stats), N, _p(ln_cb), _p(out), N, M * N, M, N, K,
                                             Z, _stream()), "ds_gemm_ln_swapped_f16")
    return out


def gemm_batched_nt(a: Tensor, b: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """out[z] = a[z or shared] @ b[z].T ; a: [M,K] or [Z,M,K], b: [Z,N,K] -> [Z,M,N]."""
    _chk(a, b)
    Z, N, K = b.shape
    M = a.shape[-2]
    sa = 0 if a.dim() == 2 else M * K
    if out is None:
        out = torch.empty((Z, M, N), dtype=torch.float16, device=b.device)
    L = _lib.load()
    check(L.ds_gemm_f16_batched(_p(a), K, sa, _p(b), K, N * K, _p(out), N, M * N, M, N, K, Z, _stream()),
          "ds_gemm_f16_batched")
    return out


def conv3x3(x: Tensor, w: Tensor, bias: Optional[Tensor], stride: int = 1, upsample: bool = False,
            rowbias: Optional[Tensor] = None, residual: Optional[Tensor] = None,
            out_size: Optional[Tuple[int, int]] = None) -> Tensor:
    """x: [B,H,W,Cin] NHWC, w: [Cout,3,3,Cin] -> [B,Ho,Wo,Cout]; rowbias: [B,Cout] per-image bias.  `out_size` (Ho, Wo):
    nearest-resize to that size first (F.interpolate(size=...) indexing), then the conv (Upsample2D with output_size)."""
    _chk(x, w, bias, rowbias, residual)
    B, H, W, Cin = x.shape
    Cout = w.shape[0]
    L = _lib.load()
    if out_size is not None:
        Ho, Wo = int(out_size[0]), int(out_size[1])
        y = torch.empty((B, Ho, Wo, Cout), dtype=torch.float16, device=x.device)
        check(L.ds_conv3x3_resize_f16(_p(x), _p(w), _p(bias), _p(rowbias), 0 if rowbias is None else rowbias.shape[1],
                                      _p(residual), _p(y), B, H, W, Cin, Cout, Ho, Wo, _stream()), "ds_conv3x3_resize_f16")
        return y
    Ho, Wo = (2 * H, 2 * W) if upsample else ((H + stride - 1) // stride, (W + stride - 1) // stride)
    y = torch.empty((B, Ho, Wo, Cout), dtype=torch.float16, device=x.device)
    check(L.ds_conv3x3_f16(_p(x), _p(w), _p(bias), _p(rowbias), 0 if rowbias is None else rowbias.shape[1],
                           _p(residual), _p(y), B, H, W, Cin, Cout, stride, int(upsample), _stream()),
          "ds_conv3x3_f16")
    return y


def groupnorm(x: Tensor, gamma: Tensor, beta: Tensor, groups: int, eps: float, silu: bool,
              x2: Optional[Tensor] = None) -> Tensor:
    """x: [B,HW,C1] (+ x2 [B,HW,C2] concatenated on channels) -> [B,HW,C1+C2]."""
    _chk(x, gamma, beta, x2)
    B, HW, C1 = x.shape
    C2 = 0 if x2 is None else x2.shape[2]
    L = _lib.load()
    ws = torch.empty(L.ds_groupnorm_workspace_bytes(B, C1 + C2), dtype=torch.uint8, device=x.device)
    y = torch.empty((B, HW, C1 + C2), dtype=torch.float16, device=x.device)
    check(L.ds_groupnorm_f16(_p(x), _p(x2), _p(y), _p(gamma), _p(beta), _p(ws), B, HW, C1, C2, groups, eps, int(silu),
                             _stream()), "ds_groupnorm_f16")
    return y


def layernorm(x: Tensor, gamma: Tensor, beta: Tensor, eps: float = 1e-5) -> Tensor:
    _chk(x, gamma, beta)
    C = x.shape[-1]
    y = torch.empty_like(x)
    check(_lib.load().ds_layernorm_f16(_p(x), _p(y), _p(gamma), _p(beta), x.numel() // C, C, eps, _stream()),
          "ds_layernorm_f16")
    return y


def self_attention(q: Tensor, k: Tensor, vt: Tensor, heads: int, scale: Optional[float] = None) -> Tensor:
    """q,k: [B,N,heads*64]; vt: [B,heads,64,Nk] (V transposed) -> [B,N,heads*64]."""
    _chk(q, k, vt)
    B, Nq, Cc = q.shape
    Nk = k.shape[1]
    o = torch.empty_like(q)
    check(_lib.load().ds_self_attn_f16(_p(q), Cc, Nq * Cc, _p(k), Cc, Nk * Cc, _p(vt), vt.shape[-1], _p(o), Cc, Nq * Cc,
                                       B, heads, Nq, Nk, scale if scale is not None else 0.125, _stream()),
          "ds_self_attn_f16")
    return o


def masked_ip_attention(q: Tensor, kt: Tensor, vtt: Tensor, ki: Tensor, vti: Tensor, bbox: Tensor, heads: int,
                        mask_hw: Tuple[int, int], ip_scale: float, Lt: int = 77, Li: int = 80, n_dummy: int = 16,
                        tok_per_ip: int = 16, qk_scale: float = 0.125) -> Tensor:
    """q: [B,N,C]; kt/ki: [B,96,C]; vtt/vti: [B,C,96]; bbox: [B,max_ips,4] fp32."""
    _chk(q, kt, vtt, ki, vti)
    _chk(bbox, dtype=torch.float32)
    B, N, Cc = q.shape
    o = torch.empty_like(q)
    check(_lib.load().ds_masked_ip_attn_f16(_p(q), Cc, _p(kt), _p(vtt), _p(ki), _p(vti), _p(bbox), _p(o), Cc, B, heads,
                                            N, Lt, Li, n_dummy, tok_per_ip, bbox.shape[1], mask_hw[0], mask_hw[1],
                                            qk_scale, ip_scale, None, 0, 0, 0, _stream()), "ds_masked_ip_attn_f16")
    return o


def ip_region_flags(bbox: Tensor, N: int, mask_hw: Tuple[int, int]) -> Tensor:
    _chk(bbox, dtype=torch.float32)
    B = bbox.shape[0]
    flags = torch.empty((B, N), dtype=torch.uint8, device=bbox.device)
    check(_lib.load().ds_ip_region_flags(_p(bbox), _p(flags), B, N, bbox.shape[1], mask_hw[0], mask_hw[1], _stream()),
          "ds_ip_region_flags")
    return flags


def embed_tokens(ids: Tensor, tok_emb: Tensor, pos_emb: Tensor) -> Tensor:
    """ids: int32 [B,T]; tok_emb: [vocab,D]; pos_emb: [>=T,D] -> [B,T,D] = tok_emb[ids] + pos_emb[:T]."""
    _chk(tok_emb, pos_emb)
    _chk(ids, dtype=torch.int32)
    B, T = ids.shape
    D = tok_emb.shape[1]
    out = torch.empty((B, T, D), dtype=torch.float16, device=ids.device)
    check(_lib.load().ds_embed_tokens_f16(_p(ids), _p(tok_emb), _p(pos_emb), _p(out), B, T, D, tok_emb.shape[0],
                                          _stream()), "ds_embed_tokens_f16")
    return out


def causal_attention(q: Tensor, k: Tensor, v: Tensor, heads: int, scale: float) -> Tensor:
    """Causal self-attention over [B,N,heads*D] (k/v may be column-slice views): the CLIP text encoders."""
    _chk(q)
    for t in (k, v):
        if not t.is_cuda or t.dtype != torch.float16 or t.stride(2) != 1:
            raise _lib.DiffSenseiHipError("causal_attention: k/v must be fp16 device tensors with unit inner stride")
    B, N, Cc = q.shape
    o = torch.empty_like(q)
    check(_lib.load().ds_small_attn_causal_f16(_p(q), Cc, N * Cc, _p(k), k.stride(1), k.stride(0), _p(v), v.stride(1),
                                               v.stride(0), _p(o), Cc, N * Cc, B, heads, N, Cc // heads, scale,
                                               _stream()), "ds_small_attn_causal_f16")
    return o


def small_attention(q: Tensor, k: Tensor, v: Tensor, heads: int, scale: float) -> Tensor:
    """q: [B,Nq,heads*D], k/v: [B,Nk,heads*D] (any D<=256 multiple of 8); k/v may be column-slice views."""
    _chk(q)
    for t in (k, v):
        if not t.is_cuda or t.dtype != torch.float16 or t.stride(2) != 1:
            raise _lib.DiffSenseiHipError("small_attention: k/v must be fp16 device tensors with unit inner stride")
    B, Nq, Cc = q.shape
    Nk = k.shape[1]
    o = torch.empty_like(q)
    check(_lib.load().ds_small_attn_f16(_p(q), Cc, Nq * Cc, _p(k), k.stride(1), k.stride(0), _p(v), v.stride(1),
                                        v.stride(0), _p(o), Cc, Nq * Cc, B, heads, Nq, Nk, Cc // heads, scale,
                                        _stream()), "ds_small_attn_f16")
    return o


def conv_in_dialog(x: Tensor, w: Tensor, bias: Tensor, boxes: Optional[Tensor], demb: Optional[Tensor]) -> Tensor:
    """x: [B,H,W,4]; w: [Cout,3,3,4]; boxes: int32 [B,nd,4] pixel boxes or None."""
    _chk(x, w, bias, demb)
    B, H, W, Cin = x.shape
    Cout = w.shape[0]
    nd = 0 if boxes is None else boxes.shape[1]
    if boxes is not None:
        _chk(boxes, dtype=torch.int32)
    y = torch.empty((B, H, W, Cout), dtype=torch.float16, device=x.device)
    check(_lib.load().ds_conv_in_dialog_f16(_p(x), _p(w), _p(bias), _p(boxes), _p(demb), _p(y), B, H, W, Cin, Cout, nd,
                                            _stream()), "ds_conv_in_dialog_f16")
    return y


def conv_out(x: Tensor, w: Tensor, bias: Tensor) -> Tensor:
    _chk(x, w, bias)
    B, H, W, Cin = x.shape
    Cout = w.shape[0]
    y = torch.empty((B, H, W, Cout), dtype=torch.float16, device=x.device)
    check(_lib.load().ds_conv_out_f16(_p(x), _p(w), _p(bias), _p(y), B, H, W, Cin, Cout, _stream()), "ds_conv_out_f16")
    return y


def skinny_linear(x: Tensor, w: Tensor, bias: Optional[Tensor] = None, addend: Optional[Tensor] = None,
                  silu_in: bool = False, silu_out: bool = False) -> Tensor:
    _chk(x, w, bias, addend)
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty((M, N), dtype=torch.float16, device=x.device)
    check(_lib.load().ds_skinny_linear_f16(_p(x), _p(w), _p(bias), _p(addend), _p(y), M, N, K, int(silu_in),
                                           int(silu_out), _stream()), "ds_skinny_linear_f16")
    return y


def timestep_embed(table: Tensor, B: int, dim: int, flip: bool = True, freq_shift: float = 0.0,
                   ctr: Optional[Tensor] = None) -> Tensor:
    _chk(table, dtype=torch.float32)
    out = torch.empty((B, dim), dtype=torch.float16, device=table.device)
    check(_lib.load().ds_timestep_embed_f16(_p(table), _p(ctr), _p(out), B, dim, int(flip), freq_shift, _stream()),
          "ds_timestep_embed_f16")
    return out


def add_time_ids(text_embeds: Tensor, time_ids: Tensor, dim: int, flip: bool = True, freq_shift: float = 0.0) -> Tensor:
    _chk(text_embeds, time_ids)
    B, P = text_embeds.shape
    n = time_ids.shape[1]
    out = torch.empty((B, P + n * dim), dtype=torch.float16, device=text_embeds.device)
    check(_lib.load().ds_add_time_ids_f16(_p(text_embeds), _p(time_ids), _p(out), B, P, n, dim, int(flip), freq_shift,
                                          _stream()), "ds_add_time_ids_f16")
    return out


def cfg_sampler_step(eps: Tensor, latents: Tensor, model_in: Tensor, table: Tensor, kind: int, do_cfg: bool = True,
                     ctr: Optional[Tensor] = None) -> None:
    """eps: [2ns,HW,4] NHWC; latents: [ns,4,H,W] NCHW (in place); model_in: [2ns,HW,4]."""
    _chk(eps, latents, model_in)
    _chk(table, dtype=torch.float32)
    ns = latents.shape[0]
    HW = latents.shape[2] * latents.shape[3]
    check(_lib.load().ds_cfg_sampler_step_f16(_p(eps), _p(latents), _p(model_in), _p(table), _p(ctr), ns, HW, kind,
                                              int(do_cfg), _stream()), "ds_cfg_sampler_step_f16")


def prepare_model_input(latents: Tensor, model_in: Tensor, table: Tensor, do_cfg: bool = True,
                        ctr: Optional[Tensor] = None) -> None:
    _chk(latents, model_in)
    ns = latents.shape[0]
    HW = latents.shape[2] * latents.shape[3]
    check(_lib.load().ds_prepare_model_input_f16(_p(latents), _p(model_in), _p(table), _p(ctr), ns, HW, int(do_cfg),
                                                 _stream()), "ds_prepare_model_input_f16")


def nhwc_to_nchw(x: Tensor) -> Tensor:
    """[B,HW,C] -> [B,C,HW]"""
    _chk(x)
    B, HW, Cc = x.shape
    y = torch.empty((B, Cc, HW), dtype=torch.float16, device=x.device)
    check(_lib.load().ds_nhwc_to_nchw_f16(_p(x), _p(y), B, HW, Cc, _stream()), "ds_nhwc_to_nchw_f16")
    return y


def nchw_to_nhwc(x: Tensor) -> Tensor:
    """[B,C,HW] -> [B,HW,C]"""
    _chk(x)
    B, Cc, HW = x.shape
    y = torch.empty((B, HW, Cc), dtype=torch.float16, device=x.device)
    check(_lib.load().ds_nchw_to_nhwc_f16(_p(x), _p(y), B, HW, Cc, _stream()), "ds_nchw_to_nhwc_f16")
    return y


def pad_rows(x: Tensor, row_off: int, rows_in: int, rows_out: int) -> Tensor:
    """x: [B,total_rows,C] -> [B,rows_out,C] = x[:, row_off:row_off+rows_in] zero-padded."""
    _chk(x)
    B, T, Cc = x.shape
    y = torch.empty((B, rows_out, Cc), dtype=torch.float16, device=x.device)
    check(_lib.load().ds_pad_rows_f16(_p(x), _p(y), B, rows_in, rows_out, row_off, T, Cc, _stream()), "ds_pad_rows_f16")
    return y


# ------------------------------------------------------------------------------------------------ bf16 (VAE decoder)
_BF = torch.bfloat16


def conv3x3_bf16(x: Tensor, w: Tensor, bias: Tensor, upsample: bool = False, residual: Optional[Tensor] = None) -> Tensor:
    """x: [B,H,W,Cin] bf16 NHWC, w: [Cout,3,3,Cin] -> [B,Ho,Wo,Cout] (stride 1; upsample = nearest x2 in front)."""
    _chk(x, w, bias, residual, dtype=_BF)
    B, H, W, Cin = x.shape
    Cout = w.shape[0]
    Ho, Wo = (2 * H, 2 * W) if upsample else (H, W)
    y = torch.empty((B, Ho, Wo, Cout), dtype=_BF, device=x.device)
    check(_lib.load().ds_conv3x3_bf16(_p(x), _p(w), _p(bias), _p(residual), _p(y), B, H, W, Cin, Cout, int(upsample),
                                      _stream()), "ds_conv3x3_bf16")
    return y


def gemm_bf16(x: Tensor, w: Tensor, bias: Optional[Tensor] = None, residual: Optional[Tensor] = None) -> Tensor:
    """y = x @ w.T + bias (+ residual), all bf16; M, N multiples of 16, K of 128."""
    _chk(x, w, bias, residual, dtype=_BF)
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty((M, N), dtype=_BF, device=x.device)
    check(_lib.load().ds_gemm_bf16(_p(x), K, _p(w), K, _p(bias), _p(residual), N, _p(y), N, M, N, K, _stream()),
          "ds_gemm_bf16")
    return y


def gemm_batched_nt_bf16(a: Tensor, b: Tensor) -> Tensor:
    """out[z] = a @ b[z].T ; a: [M,K] shared, b: [Z,N,K] -> [Z,M,N] (V^T[z] = Wv @ X_z^T)."""
    _chk(a, b, dtype=_BF)
    Z, N, K = b.shape
    M = a.shape[0]
    out = torch.empty((Z, M, N), dtype=_BF, device=b.device)
    check(_lib.load().ds_gemm_bf16_batched(_p(a), K, 0, _p(b), K, N * K, _p(out), N, M * N, M, N, K, Z, _stream()),
          "ds_gemm_bf16_batched")
    return out


def groupnorm_bf16(x: Tensor, gamma: Tensor, beta: Tensor, groups: int, eps: float, silu: bool) -> Tensor:
    """x: [B,HW,C] bf16 -> GroupNorm (+SiLU)."""
    _chk(x, gamma, beta, dtype=_BF)
    B, HW, C = x.shape
    L = _lib.load()
    ws = torch.empty(L.ds_groupnorm_workspace_bytes(B, C), dtype=torch.uint8, device=x.device)
    y = torch.empty_like(x)
    check(L.ds_groupnorm_bf16(_p(x), _p(y), _p(gamma), _p(beta), _p(ws), B, HW, C, groups, eps, int(silu), _stream()),
          "ds_groupnorm_bf16")
    return y


def wide_attention_bf16(q: Tensor, k: Tensor, vt: Tensor, scale: float, n_valid: int = 0) -> Tensor:
    """Single head of dim 512: q, k [B,N,512], vt [B,512,N] -> [B,N,512]; keys >= n_valid (0 = all) are padding."""
    _chk(q, k, vt, dtype=_BF)
    B, N, D = q.shape
    assert D == 512 and vt.shape == (B, 512, N)
    o = torch.empty_like(q)
    check(_lib.load().ds_wide_attn_bf16(_p(q), _p(k), _p(vt), _p(o), B, N, int(n_valid), scale, _stream()),
          "ds_wide_attn_bf16")
    return o


def vae_conv_in(latents: Tensor, pq_w: Tensor, pq_b: Tensor, w: Tensor, bias: Tensor, scaling_factor: float) -> Tensor:
    """latents fp32 NCHW [B,4,H,W]; pq_w [4,4], pq_b [4] fp32; w [C,3,3,4], bias [C] bf16 -> [B,H,W,C] bf16."""
    _chk(latents, pq_w, pq_b, dtype=torch.float32)
    _chk(w, bias, dtype=w.dtype)
    B, _, H, W = latents.shape
    C = w.shape[0]
    y = torch.empty((B, H, W, C), dtype=w.dtype, device=latents.device)
    L = _lib.load()
    fn, name = (L.ds_vae_conv_in_bf16, "ds_vae_conv_in_bf16") if w.dtype == _BF else (L.ds_vae_conv_in_f16, "ds_vae_conv_in_f16")
    check(fn(_p(latents), _p(pq_w), _p(pq_b), _p(w), _p(bias), _p(y), B, H, W, C, scaling_factor, _stream()), name)
    return y


def vae_conv_out(x: Tensor, w: Tensor, bias: Tensor, denormalize: bool = False) -> Tensor:
    """x [B,H,W,C] bf16 (or f16), w [3,3,3,C], bias [3] same dtype -> image fp32 NCHW [B,3,H,W] (denormalize: (y/2+0.5).clamp(0,1))."""
    _chk(x, w, bias, dtype=x.dtype)
    B, H, W, C = x.shape
    img = torch.empty((B, 3, H, W), dtype=torch.float32, device=x.device)
    L = _lib.load()
    fn, name = (L.ds_vae_conv_out_bf16, "ds_vae_conv_out_bf16") if x.dtype == _BF else (L.ds_vae_conv_out_f16, "ds_vae_conv_out_f16")
    check(fn(_p(x), _p(w), _p(bias), _p(img), B, H, W, C, int(denormalize), _stream()), name)
    return img


# ---- f16 twins used by the decoder's scaled-fp16 mode (vae.py)
def groupnorm_scaled(x: Tensor, gamma: Tensor, beta: Tensor, groups: int, eps: float, silu: bool, out_scale: float) -> Tensor:
    """x: [B,HW,C] f16 -> act(GroupNorm(x)) * out_scale."""
    _chk(x, gamma, beta)
    B, HW, C = x.shape
    L = _lib.load()
    ws = torch.empty(L.ds_groupnorm_workspace_bytes(B, C), dtype=torch.uint8, device=x.device)
    y = torch.empty_like(x)
    check(L.ds_groupnorm_scaled_f16(_p(x), _p(y), _p(gamma), _p(beta), _p(ws), B, HW, C, groups, eps, int(silu),
                                    float(out_scale), _stream()), "ds_groupnorm_scaled_f16")
    return y


def wide_attention_f16(q: Tensor, k: Tensor, vt: Tensor, scale: float, n_valid: int = 0) -> Tensor:
    """`wide_attention_bf16` on f16 tensors."""
    _chk(q, k, vt)
    B, N, D = q.shape
    assert D == 512 and vt.shape == (B, 512, N)
    o = torch.empty_like(q)
    check(_lib.load().ds_wide_attn_f16(_p(q), _p(k), _p(vt), _p(o), B, N, int(n_valid), scale, _stream()), "ds_wide_attn_f16")
    return o


# ---- MLLM pre-pass: LLaMA greedy decoding (csrc/llm.hip) ----------------------------------------------------
def llm_gemv(x: Tensor, w: Tensor, out: Optional[Tensor] = None, residual: Optional[Tensor] = None, rms: bool = False,
             swiglu: bool = False, eps: float = 1e-6, gain: Optional[Tensor] = None) -> Tensor:
    """x: [M<=any, K] rows; w: [N,K] (swiglu: [2N,K] gate rows then up rows).  `rms`: a LlamaRMSNorm in front - with
    `gain` [K] the reference's roundings (f16(gain * f16(x / rms))), without it only the 1/rms scale (gain folded into w by
    the caller).  `residual` may be `out` itself (in-place h += ...)."""
    _chk(x, w, residual, gain)
    M, K = x.shape
    N = w.shape[0] // (2 if swiglu else 1)
    assert w.shape[1] == K
    if out is None:
        out = torch.empty((M, N), dtype=torch.float16, device=x.device)
    _chk(out)
    check(_lib.load().ds_llm_gemv_f16(_p(x), K, _p(w), _p(out), N, _p(residual), N, M, N, K, int(rms), _p(gain),
                                      int(swiglu), eps, _stream()), "ds_llm_gemv_f16")
    return out


def llm_attention(qkv: Tensor, k_cache: Tensor, v_cache: Tensor, rope_cos: Tensor, rope_sin: Tensor, state: Tensor,
                  heads: int, kv_heads: int, scale: float, out: Optional[Tensor] = None) -> Tensor:
    """qkv: [M,(heads+2*kv_heads)*D] un-rotated; caches [T_max, kv_heads*D]; appends the M rows at state[0].."""
    _chk(qkv, k_cache, v_cache)
    _chk(rope_cos, rope_sin, dtype=torch.float32)
    _chk(state, dtype=torch.int32)
    M = qkv.shape[0]
    D = qkv.shape[1] // (heads + 2 * kv_heads)
    T_max = k_cache.shape[0]
    assert rope_cos.shape == (T_max, D // 2) and k_cache.shape[1] == kv_heads * D
    if out is None:
        out = torch.empty((M, heads * D), dtype=torch.float16, device=qkv.device)
    check(_lib.load().ds_llm_attn_f16(_p(qkv), qkv.shape[1], _p(k_cache), _p(v_cache), k_cache.shape[1], _p(rope_cos),
                                      _p(rope_sin), _p(out), heads * D, _p(state), M, heads, kv_heads, D, T_max, scale,
                                      _stream()), "ds_llm_attn_f16")
    return out


def llm_rmsnorm(x: Tensor, gamma: Tensor, eps: float, out: Optional[Tensor] = None, feat: Optional[Tensor] = None,
                state: Optional[Tensor] = None) -> Tensor:
    _chk(x, gamma, feat)
    M, H = x.shape
    if out is None:
        out = torch.empty_like(x)
    check(_lib.load().ds_llm_rmsnorm_f16(_p(x), H, _p(gamma), _p(out), H, _p(feat), _p(state), M, H,
                                         0 if feat is None else feat.shape[0], eps, _stream()), "ds_llm_rmsnorm_f16")
    return out


def llm_embed(table: Tensor, state: Tensor, out: Tensor) -> Tensor:
    _chk(table, out)
    _chk(state, dtype=torch.int32)
    check(_lib.load().ds_llm_embed_f16(_p(table), _p(state), _p(out), table.shape[1], table.shape[0], _stream()),
          "ds_llm_embed_f16")
    return out


def llm_select(logits: Tensor, chain: Optional[Tensor], adv: int, state: Tensor, out_ids: Tensor) -> None:
    """Greedy pick + image-token logits processor + bookkeeping in the device state block (see the header)."""
    _chk(logits)
    _chk(chain, state, out_ids, dtype=torch.int32)
    check(_lib.load().ds_llm_select_f16(_p(logits), logits.numel(), _p(chain), 0 if chain is None else chain.numel(),
                                        out_ids.numel(), adv, _p(state), _p(out_ids), _stream()), "ds_llm_select_f16")


def llm_advance(state: Tensor, rows: int) -> None:
    _chk(state, dtype=torch.int32)
    check(_lib.load().ds_llm_advance(_p(state), rows, _stream()), "ds_llm_advance")


def blend(a: Tensor, b: Tensor, scale: float) -> Tensor:
    """a*scale + b*(1-scale) (fp16, same shape, numel % 8 == 0)."""
    _chk(a, b)
    assert a.shape == b.shape
    out = torch.empty_like(a)
    check(_lib.load().ds_blend_f16(_p(a), _p(b), _p(out), a.numel(), scale, _stream()), "ds_blend_f16")
    return out


def llm_swiglu(gate_up: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """[M, 2I] (gate | up) -> silu(gate) * up  [M, I]."""
    _chk(gate_up, out)
    M, I2 = gate_up.shape
    if out is None:
        out = torch.empty((M, I2 // 2), dtype=torch.float16, device=gate_up.device)
    check(_lib.load().ds_llm_swiglu_f16(_p(gate_up), _p(out), M, I2 // 2, _stream()), "ds_llm_swiglu_f16")
    return out


def image_to_u8(image: Tensor) -> Tensor:
    """[B,3,H,W] fp32 in [0,1] -> [B,H,W,3] uint8 = (x*255).round() (half to even), the PIL tail of `postprocess`."""
    _chk(image, dtype=torch.float32)
    B, Cc, H, W = image.shape
    if Cc != 3:
        raise _lib.DiffSenseiHipError("image_to_u8: 3 channels expected")
    out = torch.empty((B, H, W, 3), dtype=torch.uint8, device=image.device)
    check(_lib.load().ds_image_f32_to_u8_nhwc(_p(image), _p(out), B, H, W, _stream()), "ds_image_f32_to_u8_nhwc")
    return out
