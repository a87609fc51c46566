"""UNet launch-plan builder: packs the diffusers-layout state dict for the HIP kernels and emits the flat
op list (`ds_op[]`) that the C++ plan executor replays for one `UNetMangaModel.forward`
(reference src/models/unet.py:185-338) and, for sampling, forward + CFG + scheduler step
(reference src/pipelines/pipeline_diffsensei.py:310-337).

Host work happens ONCE per (batch, height, width) bucket; the steady state is `ds_plan_run/replay`.
Per-request constants are hoisted out of the step loop (the reference recomputes them every layer of every
step): text/IP key and value projections for all cross-attention layers (4 stacked GEMMs), the `text_time`
added-condition embedding, and the region-mask geometry.

HBM layout: activations fp16 channels-last [B*H*W, C] (a conv output IS the token matrix of the following
transformer); conv weights [Cout, 3*3*Cin]; V operands of attention stored key-contiguous.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib
from ._lib import OP, DsOp, check
from .attention_processor import mask_grid_size
from .unet_config import AttnSpec, ResnetSpec, UNetMangaConfig, build_topology

Tensor = torch.Tensor
LP = 96  # padded key-panel length of the fused cross-attention kernel


# ------------------------------------------------------------------------------------------ weight packing
def pack_conv3x3(w: Tensor) -> Tensor:
    """[Cout,Cin,3,3] -> [Cout, 9*Cin] with k = (ky*3+kx)*Cin + ci (matches the NHWC implicit-GEMM gather)."""
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()


def pack_geglu(w: Tensor, b: Tensor) -> Tuple[Tensor, Tensor]:
    """GEGLU proj [8C,C]: rows [0,4C) 'hidden', [4C,8C) 'gate' -> per 128-row tile 64 hidden rows + their 64 gate rows."""
    half = w.shape[0] // 2
    assert half % 64 == 0, "GEGLU inner dim must be a multiple of 64"
    t = half // 64
    wp = torch.stack([w[:half].reshape(t, 64, -1), w[half:].reshape(t, 64, -1)], dim=1).reshape(2 * half, -1)
    bp = torch.stack([b[:half].reshape(t, 64), b[half:].reshape(t, 64)], dim=1).reshape(2 * half)
    return wp.contiguous(), bp.contiguous()


def unpack_geglu(t: Tensor) -> Tensor:
    """Inverse of `pack_geglu` on its first axis: 128-row groups (64 hidden + 64 gates) -> [hidden rows | gate rows]."""
    g = t.reshape(t.shape[0] // 128, 2, 64, *t.shape[1:])
    return torch.cat([g[:, 0].reshape(-1, *t.shape[1:]), g[:, 1].reshape(-1, *t.shape[1:])], 0)


def pack_geglu320(t: Tensor) -> Tensor:
    """GEGLU rows [hidden | gate] (weight [8C,C], bias [8C] or the fused-LayerNorm c pairs [8C,2]) -> groups of 320 rows: 160
    hidden rows followed by their 160 gate rows - the layout gemm_g320_kernel's 256 x 320 tiles read (csrc/gemm_g320.hip)."""
    half = t.shape[0] // 2
    assert half % 160 == 0, "GEGLU inner dim must be a multiple of 160"
    g = half // 160
    return torch.stack([t[:half].reshape(g, 160, *t.shape[1:]), t[half:].reshape(g, 160, *t.shape[1:])], dim=1).reshape(t.shape).contiguous()


def pack_ln_fused(w: Tensor, bias: Optional[Tensor], gamma: Tensor, beta: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """LayerNorm folded into the linear that consumes it (csrc/gemm_pp.hip, "LayerNorm"):
    LN(x) W^T + b = rstd (x (gamma (.) W)^T - mean c) + b',  c_n = sum_k (gamma (.) W)_nk,  b' = b + W beta.
    Returns (gw f16 [N,K], c2 f16 [N,2] = (-c hi, -c lo), b' f16 [N]).  c is summed from the f16-ROUNDED gw - the values
    the MFMA contracts - so the rank-1 correction cancels the mean exactly; the (hi, lo) pair keeps it to 2^-22."""
    wf = w.double()
    gw = (wf * gamma.double()[None, :]).to(torch.float16)
    nc = (-gw.double().sum(1)).float()
    hi = nc.to(torch.float16)
    lo = (nc - hi.float()).to(torch.float16)
    bp = wf @ beta.double()
    if bias is not None:
        bp = bp + bias.double()
    return gw.contiguous(), torch.stack([hi, lo], dim=1).contiguous(), bp.to(torch.float16).contiguous()


def gn_fusion_enabled() -> bool:
    """DIFFSENSEI_GN_FUSION=0 keeps the three-launch GroupNorm behind conv1 of every resnet (A/B runs): by default conv1's
    epilogue emits the partial sums norm2 needs (csrc/conv_halo.hip) and the GroupNorm reads its input once."""
    import os
    return os.environ.get("DIFFSENSEI_GN_FUSION", "1") != "0"


def ln_fusion_enabled() -> bool:
    """DIFFSENSEI_LN_FUSION=0 keeps the stand-alone LayerNorm launches everywhere (A/B runs)."""
    import os
    return os.environ.get("DIFFSENSEI_LN_FUSION", "1") != "0"


# Where the fused LayerNorm pays (tools/ln_fusion_sweep.py, profiles/r04_ln_fusion_sweep.txt: every UNet batch 2..64, each
# kernel family alone, both, twice on two boxes): fusing every level whose GEMMs implement it wins at every batch - 0.9 ms of
# 32.4 at batch 2 (963 -> 763 launches), 1.3 of 73.5 at batch 8, 2.2 of 132 at 16, 4.2 of 242 at 32, 9.5 of 467 at 64 - and
# no family alone does better anywhere.  The three limits (in elements = rows x channels of the normalised tensor) therefore
# default to "no limit"; they exist for the sweep.
LN_FUSION_ALL_PP_MIN_ELEMS = 0


def ln_fusion_limits() -> Tuple[int, int, int]:
    """(smallest tensor a gemm_pp_kernel member may fuse at, largest a 128-wide member may, smallest for an all-gemm_pp level)"""
    import os
    return (int(os.environ.get("DIFFSENSEI_LN_FUSION_PP_MIN_ELEMS", 0)),
            int(os.environ.get("DIFFSENSEI_LN_FUSION_WIDE_MAX_ELEMS", 1 << 62)),
            int(os.environ.get("DIFFSENSEI_LN_FUSION_ALL_PP_MIN_ELEMS", LN_FUSION_ALL_PP_MIN_ELEMS)))


class PackedUNet:
    """Device-resident fp16 weights in kernel layout.  `names` follow the diffusers state dict."""

    def __init__(self, cfg: UNetMangaConfig, sd: Dict[str, Tensor], device: torch.device):
        self.cfg = cfg
        self.device = device
        self.topo = build_topology(cfg)
        self.w: Dict[str, Tensor] = {}
        f16 = lambda t: t.detach().to(device=device, dtype=torch.float16).contiguous()

        def put(name, t):
            self.w[name] = f16(t)

        resnets: List[ResnetSpec] = []
        attns: List[AttnSpec] = []
        for blk in self.topo.down:
            resnets += blk["resnets"]
            attns += blk["attns"]
        resnets += self.topo.mid["resnets"]
        attns += self.topo.mid["attns"]
        for blk in self.topo.up:
            resnets += blk["resnets"]
            attns += blk["attns"]
        self.resnets, self.attns = resnets, attns

        put("conv_in.weight", pack_conv3x3(sd["conv_in.weight"]))
        put("conv_in.bias", sd["conv_in.bias"])
        put("conv_out.weight", pack_conv3x3(sd["conv_out.weight"]))
        put("conv_out.bias", sd["conv_out.bias"])
        for n in ("conv_norm_out.weight", "conv_norm_out.bias", "dialog_bbox_embedding",
                  "time_embedding.linear_1.weight", "time_embedding.linear_1.bias",
                  "time_embedding.linear_2.weight", "time_embedding.linear_2.bias",
                  "add_embedding.linear_1.weight", "add_embedding.linear_1.bias",
                  "add_embedding.linear_2.weight", "add_embedding.linear_2.bias"):
            put(n, sd[n])
        # resnets; all time_emb_proj stacked into one [sum Cout, temb] matrix (one skinny GEMM per forward)
        tw, tb, off = [], [], 0
        self.temb_off: Dict[str, int] = {}
        for r in resnets:
            p = r.prefix
            for n in ("norm1", "norm2"):
                put(f"{p}.{n}.weight", sd[f"{p}.{n}.weight"])
                put(f"{p}.{n}.bias", sd[f"{p}.{n}.bias"])
            put(f"{p}.conv1.weight", pack_conv3x3(sd[f"{p}.conv1.weight"]))
            put(f"{p}.conv1.bias", sd[f"{p}.conv1.bias"])
            put(f"{p}.conv2.weight", pack_conv3x3(sd[f"{p}.conv2.weight"]))
            put(f"{p}.conv2.bias", sd[f"{p}.conv2.bias"])
            if r.has_shortcut:
                put(f"{p}.conv_shortcut.weight", sd[f"{p}.conv_shortcut.weight"].reshape(r.cout, r.cin))
                put(f"{p}.conv_shortcut.bias", sd[f"{p}.conv_shortcut.bias"])
            tw.append(sd[f"{p}.time_emb_proj.weight"])
            tb.append(sd[f"{p}.time_emb_proj.bias"])
            self.temb_off[p] = off
            off += r.cout
        self.temb_total = off
        put("temb_all.weight", torch.cat([t.to(device) for t in tw], 0))
        put("temb_all.bias", torch.cat([t.to(device) for t in tb], 0))
        for blk in self.topo.down:
            if blk["downsample"]:
                put(blk["downsample"] + ".weight", pack_conv3x3(sd[blk["downsample"] + ".weight"]))
                put(blk["downsample"] + ".bias", sd[blk["downsample"] + ".bias"])
        for blk in self.topo.up:
            if blk["upsample"]:
                put(blk["upsample"] + ".weight", pack_conv3x3(sd[blk["upsample"] + ".weight"]))
                put(blk["upsample"] + ".bias", sd[blk["upsample"] + ".bias"])
        # transformers; cross-attention K/V projections of ALL layers stacked (hoisted out of the step loop)
        kt, vt, ki, vi, off = [], [], [], [], 0
        self.kv_off: Dict[str, int] = {}
        for a in attns:
            p = a.prefix
            for n in ("norm.weight", "norm.bias", "proj_in.weight", "proj_in.bias", "proj_out.weight", "proj_out.bias"):
                put(f"{p}.{n}", sd[f"{p}.{n}"])
            for k in range(a.depth):
                t = f"{p}.transformer_blocks.{k}"
                for n in ("norm1", "norm2", "norm3"):
                    put(f"{t}.{n}.weight", sd[f"{t}.{n}.weight"])
                    put(f"{t}.{n}.bias", sd[f"{t}.{n}.bias"])
                put(f"{t}.attn1.qk.weight", torch.cat([sd[f"{t}.attn1.to_q.weight"].to(device),
                                                       sd[f"{t}.attn1.to_k.weight"].to(device)], 0))
                put(f"{t}.attn1.to_v.weight", sd[f"{t}.attn1.to_v.weight"])
                put(f"{t}.attn1.to_out.0.weight", sd[f"{t}.attn1.to_out.0.weight"])
                put(f"{t}.attn1.to_out.0.bias", sd[f"{t}.attn1.to_out.0.bias"])
                put(f"{t}.attn2.to_q.weight", sd[f"{t}.attn2.to_q.weight"])
                put(f"{t}.attn2.to_out.0.weight", sd[f"{t}.attn2.to_out.0.weight"])
                put(f"{t}.attn2.to_out.0.bias", sd[f"{t}.attn2.to_out.0.bias"])
                kt.append(sd[f"{t}.attn2.to_k.weight"])
                vt.append(sd[f"{t}.attn2.to_v.weight"])
                ki.append(sd[f"{t}.attn2.processor.to_k_ip.weight"])
                vi.append(sd[f"{t}.attn2.processor.to_v_ip.weight"])
                self.kv_off[t] = off
                off += a.channels
                wp, bp = pack_geglu(sd[f"{t}.ff.net.0.proj.weight"].to(device), sd[f"{t}.ff.net.0.proj.bias"].to(device))
                put(f"{t}.ff.net.0.proj.weight", wp)
                put(f"{t}.ff.net.0.proj.bias", bp)
                dv = lambda n: sd[n].to(device)
                if a.channels % 128 == 0 and ln_fusion_enabled():
                    # (with DIFFSENSEI_LN_FUSION=0 in the environment at LOAD time the +2.1 GB of copies are not made
                    # at all; a plan built later simply finds none and keeps the LayerNorm launches.  With the default - on -
                    # they are packed once here and the plan builder decides per level and batch which set it uses.)
                    # fused-LayerNorm copies for norm1 -> q|k (row form) and -> to_v, produced transposed (operand-swapped form:
                    # (-c, b') per output row)
                    gw, c2, b2 = pack_ln_fused(self.w[f"{t}.attn1.qk.weight"], None, dv(f"{t}.norm1.weight"), dv(f"{t}.norm1.bias"))
                    put(f"{t}.attn1.qk.weight_ln", gw)
                    put(f"{t}.attn1.qk.c_ln", c2)
                    put(f"{t}.attn1.qk.bias_ln", b2)
                    gw, c2, b2 = pack_ln_fused(dv(f"{t}.attn1.to_v.weight"), None, dv(f"{t}.norm1.weight"), dv(f"{t}.norm1.bias"))
                    bf = (dv(f"{t}.attn1.to_v.weight").double() @ dv(f"{t}.norm1.bias").double()).float()   # b' as an f16 (hi, lo) pair
                    bh = bf.to(torch.float16)
                    bl = (bf - bh.float()).to(torch.float16)
                    put(f"{t}.attn1.to_v.weight_ln", gw)
                    put(f"{t}.attn1.to_v.cb_ln", torch.cat([c2, torch.stack([bh, bl], dim=1)], dim=1).contiguous())
                    # norm2 -> attn2.to_q, norm3 -> GEGLU projection.  Both kernel families implement every form (the 128-wide
                    # producers need whole 64-column strips in a 128-column tile), so every SDXL block gets the copies (+2.1 GB)
                    gw, c2, b2 = pack_ln_fused(dv(f"{t}.attn2.to_q.weight"), None, dv(f"{t}.norm2.weight"), dv(f"{t}.norm2.bias"))
                    put(f"{t}.attn2.to_q.weight_ln", gw)
                    put(f"{t}.attn2.to_q.c_ln", c2)
                    put(f"{t}.attn2.to_q.bias_ln", b2)
                    gw, c2, b2 = pack_ln_fused(dv(f"{t}.ff.net.0.proj.weight"), dv(f"{t}.ff.net.0.proj.bias"),
                                               dv(f"{t}.norm3.weight"), dv(f"{t}.norm3.bias"))
                    gwp, b2p = pack_geglu(gw, b2)
                    half = c2.shape[0] // 2
                    c2p = torch.stack([c2[:half].reshape(-1, 64, 2), c2[half:].reshape(-1, 64, 2)], dim=1).reshape(-1, 2)
                    put(f"{t}.ff.net.0.proj.weight_ln", gwp)
                    put(f"{t}.ff.net.0.proj.c_ln", c2p)
                    put(f"{t}.ff.net.0.proj.bias_ln", b2p)
                put(f"{t}.ff.net.2.weight", sd[f"{t}.ff.net.2.weight"])
                put(f"{t}.ff.net.2.bias", sd[f"{t}.ff.net.2.bias"])
        self.kv_total = off
        for name, lst in (("xattn.k_text", kt), ("xattn.v_text", vt), ("xattn.k_ip", ki), ("xattn.v_ip", vi)):
            put(name, torch.cat([t.to(device) for t in lst], 0))

    def geglu320(self, t: str, ln: bool) -> Tuple[Tensor, Tensor, Optional[Tensor]]:
        """(weight, bias, c) of transformer block `t`'s GEGLU projection re-packed in 320-row groups for gemm_g320_kernel - made
        on first use (a plan asks only where ds_gemm_g320_fits says that kernel runs: the 1280-channel level of a batch-1
        request; +26 MB per block) from the 128-row packing, which stays for every other batch."""
        sfx = "_ln" if ln else ""
        key = f"{t}.ff.net.0.proj.weight{sfx}.g320"
        if key not in self.w:
            self.w[key] = pack_geglu320(unpack_geglu(self.w[f"{t}.ff.net.0.proj.weight{sfx}"]))
            self.w[f"{t}.ff.net.0.proj.bias{sfx}.g320"] = pack_geglu320(unpack_geglu(self.w[f"{t}.ff.net.0.proj.bias{sfx}"]))
            if ln:
                self.w[f"{t}.ff.net.0.proj.c_ln.g320"] = pack_geglu320(unpack_geglu(self.w[f"{t}.ff.net.0.proj.c_ln"]))
        return self.w[key], self.w[f"{t}.ff.net.0.proj.bias{sfx}.g320"], (self.w[f"{t}.ff.net.0.proj.c_ln.g320"] if ln else None)

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.w.values())

    def all_tensors(self) -> List[Tensor]:
        return list(self.w.values())


# ------------------------------------------------------------------------------------------ op helpers
def _ptr(x) -> Optional[int]:
    if x is None:
        return None
    if isinstance(x, int):
        return x
    return x.data_ptr()


def make_op(code: str, i=(), f=(), l=(), p=()) -> DsOp:
    op = DsOp()
    op.code = OP[code]
    for k, v in enumerate(i):
        op.i[k] = int(v)
    for k, v in enumerate(f):
        op.f[k] = float(v)
    for k, v in enumerate(l):
        op.l[k] = int(v)
    for k, v in enumerate(p):
        op.p[k] = _ptr(v)
    return op


class Plan:
    """Owns a ds_plan* plus every tensor whose pointer is baked into it."""

    def __init__(self, ops: List[DsOp], keep: list):
        self.lib = _lib.load()
        self.dev = next((t.device for t in keep if isinstance(t, torch.Tensor) and t.is_cuda), None)
        self.n = len(ops)
        arr = (DsOp * self.n)(*ops)
        h = C.c_void_p()
        check(self.lib.ds_plan_create(arr, self.n, C.byref(h)), "ds_plan_create")
        self.handle = h
        self.keep = keep
        self.captured = False

    def _bind(self):
        if self.dev is not None and self.dev.index is not None and self.dev.index != torch.cuda.current_device():
            torch.cuda.set_device(self.dev)   # launches resolve "the current stream" on the plan's own GPU

    def run(self, stream: Optional[int] = None):
        self._bind()
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        check(self.lib.ds_plan_run(self.handle, s), "ds_plan_run")

    def capture(self, stream: int):
        check(self.lib.ds_plan_capture(self.handle, stream), "ds_plan_capture")
        self.captured = True

    def replay(self, stream: Optional[int] = None):
        self._bind()
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        check(self.lib.ds_plan_replay(self.handle, s), "ds_plan_replay")

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.ds_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


# ------------------------------------------------------------------------------------------ the engine
class UNetEngine:
    """Static-shape execution state for one (batch, latent H, latent W) bucket."""

    def __init__(self, packed: PackedUNet, batch: int, height: int, width: int, aspect_ratio: Optional[float] = None,
                 attention: str = "fp16"):
        """attention: "fp16" - the reference's arithmetic, the only one since round 6 (the OCP e4m3 variant of rounds 2-5 was
        retired: 5.4e-2 per op for +3.7 % at 2048 x 2048)."""
        if attention != "fp16":
            raise ValueError(f"attention must be 'fp16', got {attention!r} (the fp8 variant was removed in round 6)")
        self.attention = attention
        cfg = packed.cfg
        self.pk, self.cfg = packed, cfg
        self.B, self.H, self.W = batch, height, width
        self.dev = packed.device
        self.aspect_ratio = aspect_ratio if aspect_ratio is not None else height / width
        self.hw = self.level_sizes(cfg, height, width)
        self.keep: list = []
        B = batch
        E = lambda *shape, dtype=torch.float16: self._alloc(shape, dtype)
        c0 = cfg.block_out_channels[0]
        temb = cfg.time_embed_dim
        n_txt, n_ip = cfg.num_text_tokens, cfg.num_ip_tokens
        xdim = cfg.cross_attention_dim
        # ---- request-level inputs (written by the host before `prepare`)
        self.enc = E(B, n_txt + n_ip, xdim)                     # encoder_hidden_states = [text | dummy+ip]
        self.text_embeds = E(B, cfg.projection_class_embeddings_input_dim - 6 * cfg.addition_time_embed_dim)
        self.time_ids = E(B, 6)
        self.bbox = E(B, cfg.max_num_ips, 4, dtype=torch.float32)
        self.dialog_boxes = E(B, cfg.max_num_dialogs, 4, dtype=torch.int32)
        self.dialog_boxes.zero_()
        self.ip_scale = E(1, dtype=torch.float32)
        self.ip_scale.fill_(1.0)
        self.table = E(1024, 8, dtype=torch.float32)            # per-step scalars (see include/diffsensei_hip.h)
        self.table.zero_()
        self.ctr = E(1, dtype=torch.int32)
        self.ctr.zero_()
        # ---- request-level derived tensors
        self.enc_txt = E(B, LP, xdim)
        self.enc_ip = E(B, LP, xdim)
        kvt = packed.kv_total
        self.k_txt = E(B * LP, kvt)
        self.k_ip = E(B * LP, kvt)
        self.v_txt = E(B, kvt, LP)
        self.v_ip = E(B, kvt, LP)
        self.add_in = E(B, cfg.projection_class_embeddings_input_dim)
        self.add_h = E(B, temb)
        self.aug = E(B, temb)
        # ---- per-forward small tensors
        self.t_sin = E(B, c0)
        self.t_h = E(B, temb)
        self.emb = E(B, temb)
        self.temb_all = E(B, packed.temb_total)
        # ---- activations
        self.x_in = E(B, height * width, cfg.in_channels)
        self.eps = E(B, height * width, cfg.out_channels)
        self.gn_ws = self._alloc((_lib.load().ds_groupnorm_workspace_bytes(B, 4 * cfg.block_out_channels[-1]),),
                                 torch.uint8)
        self.scratch: Dict[Tuple[str, int], Tensor] = {}
        self.prepare_plan = self._build_prepare()
        self.forward_ops = self._build_forward()
        self.forward_plan = Plan(self.forward_ops, self.keep)

    @staticmethod
    def level_sizes(cfg: UNetMangaConfig, height: int, width: int) -> List[Tuple[int, int]]:
        """(h, w) of every resolution level for a latent of height x width.  Any size works, like the reference (image sides
        that are multiples of 8): a stride-2 Downsample2D gives ceil(h / 2), and the matching Upsample2D resizes to the skip
        tensor's size (diffusers `forward_upsample_size` [3P]) - see `ds_conv3x3_resize_f16`."""
        if height < 1 or width < 1:
            raise ValueError(f"latent size {height}x{width}: both sides must be >= 1")
        hw = [(int(height), int(width))]
        for _ in range(len(cfg.block_out_channels) - 1):
            h, w = hw[-1]
            hw.append(((h + 1) // 2, (w + 1) // 2))
        return hw

    # -- memory
    def _alloc(self, shape, dtype) -> Tensor:
        t = torch.empty(tuple(int(s) for s in shape), dtype=dtype, device=self.dev)
        self.keep.append(t)
        return t

    def _buf(self, role: str, level: int, rows: int, cols: int, zero: bool = False) -> Tensor:
        """Role-keyed scratch, sized to the largest request per (role, level); `zero`: cleared once at allocation (for
        buffers whose slack is read but never written)."""
        key = (role, level)
        need = rows * cols
        t = self.scratch.get(key)
        if t is None or t.numel() < need:
            t = self._alloc((need,), torch.float16)
            if zero:
                t.zero_()
            self.scratch[key] = t
        return t

    def _buf32(self, role: str, level: int, n: int) -> Tensor:
        """fp32 scratch keyed like `_buf` (row statistics of the fused LayerNorms)."""
        key = (role, level)
        t = self.scratch.get(key)
        if t is None or t.numel() < n:
            t = self._alloc((n,), torch.float32)
            self.scratch[key] = t
        return t

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.keep)

    # -- request-level plan: everything that does not depend on the timestep
    def _build_prepare(self) -> Plan:
        cfg, pk, B = self.cfg, self.pk, self.B
        w = pk.w
        n_txt, n_ip, xdim = cfg.num_text_tokens, cfg.num_ip_tokens, cfg.cross_attention_dim
        kvt = pk.kv_total
        ops = [
            make_op("PAD_ROWS", i=(B, n_txt, LP, 0, n_txt + n_ip, xdim), p=(self.enc, self.enc_txt)),
            make_op("PAD_ROWS", i=(B, n_ip, LP, n_txt, n_txt + n_ip, xdim), p=(self.enc, self.enc_ip)),
        ]
        for enc, wk, wv, kout, vout in ((self.enc_txt, w["xattn.k_text"], w["xattn.v_text"], self.k_txt, self.v_txt),
                                        (self.enc_ip, w["xattn.k_ip"], w["xattn.v_ip"], self.k_ip, self.v_ip)):
            # K panels of every layer: [B*96, sumC] = enc_pad @ Wk_all^T
            ops.append(make_op("GEMM", i=(B * LP, kvt, xdim, xdim, 0, 1, 0, 1), l=(xdim, 0, xdim, kvt, 0),
                               p=(enc, None, wk, kout)))
            # V^T panels: per image [sumC, 96] = Wv_all @ enc_pad[b]^T
            ops.append(make_op("GEMM", i=(kvt, LP, xdim, xdim, 0, B, 0, 1),
                               l=(xdim, 0, xdim, LP, 0, 0, 0, LP * xdim, kvt * LP, 0), p=(wv, None, enc, vout)))
        pooled = self.text_embeds.shape[1]
        ops += [
            make_op("ADD_TIME_IDS", i=(B, pooled, 6, cfg.addition_time_embed_dim, int(cfg.flip_sin_to_cos)),
                    f=(cfg.freq_shift,), p=(self.text_embeds, self.time_ids, self.add_in)),
            make_op("SKINNY", i=(B, cfg.time_embed_dim, self.add_in.shape[1], 0, 1),
                    p=(self.add_in, w["add_embedding.linear_1.weight"], w["add_embedding.linear_1.bias"], None, self.add_h)),
            make_op("SKINNY", i=(B, cfg.time_embed_dim, cfg.time_embed_dim, 0, 0),
                    p=(self.add_h, w["add_embedding.linear_2.weight"], w["add_embedding.linear_2.bias"], None, self.aug)),
        ]
        return Plan(ops, self.keep)

    # -- building blocks of the forward plan
    def _gn(self, ops, x1, x2, y, gamma, beta, HW, C1, C2, eps, silu, pre_chunks=0):
        """pre_chunks > 0: the producing convolution left that many partial-sum chunks per image in the workspace (`_conv(...,
        gn_stats=True)`): the GroupNorm skips its statistics launch and reads x once."""
        ops.append(make_op("GROUPNORM", i=(self.B, HW, C1, C2, self.cfg.norm_num_groups, int(silu), int(pre_chunks)), f=(eps,),
                           p=(x1, x2, y, gamma, beta, self.gn_ws)))

    def _conv(self, ops, x, wname, y, H, W, Cin, Cout, stride=1, upsample=0, rowbias=None, residual=None, out_hw=(0, 0),
              gn_stats=False):
        """gn_stats: the convolution's epilogue also writes the GroupNorm partial sums of its output into the workspace
        (csrc/conv_halo.hip; the caller has asked `ds_conv3x3_gn_chunks` first and passes the count to the GroupNorm op)."""
        w = self.pk.w
        ops.append(make_op("CONV3X3", i=(self.B, H, W, Cin, Cout, stride, upsample, self.pk.temb_total, out_hw[0], out_hw[1]),
                           p=(x, w[wname + ".weight"], y, w[wname + ".bias"], rowbias, residual, self.gn_ws if gn_stats else None)))

    def _gemm(self, ops, x, wt, y, M, N, K, bias=None, residual=None, geglu=False, x2=None, K1=0, ln_stats=None, ln_c=None,
              stats_out=None, ln_partial=False, ln_nstrips=0, stats_strip=0):
        """ln_stats / ln_c: this GEMM consumes a fused LayerNorm (x is the raw residual stream, wt / bias the `_ln` copies);
        ln_partial: ln_stats are the producer's partial sums and the kernel finalises its own rows (the 128-wide kernels);
        stats_out: it emits the row statistics of what it stores (csrc/gemm_pp.hip, "LayerNorm"; csrc/gemm.hip, "Fused LayerNorm");
        stats_strip: statistics format the producer is asked for (0 / 64 = one entry per 64 columns; 160 = gemm_t160_kernel's three per tile),
        ln_nstrips: entries per row a consumer of partial sums adds up (0 = K / 64)."""
        n_out = N // 2 if geglu else N
        epi = 4 if geglu == 320 else int(bool(geglu))   # 4: W / bias / c packed in 320-row groups (PackedUNet.geglu320)
        ops.append(make_op("GEMM", i=(M, N, K, K1 if x2 is not None else K, epi, 1, 0, 1, 0, int(ln_partial),
                                      int(ln_nstrips) if ln_stats is not None else 0, int(stats_strip) if stats_out is not None else 0),
                           f=(1e-5,),
                           l=(K1 if x2 is not None else K, (K - K1) if x2 is not None else 0, K, n_out, n_out),
                           p=(x, x2, wt, y, bias, None, residual, ln_stats, ln_c, stats_out)))

    def _resnet(self, ops, r: ResnetSpec, x1: Tensor, x2: Optional[Tensor], c1: int, c2: int, out: Tensor, in_chunks: int = 0,
                out_stats: bool = False) -> int:
        """in_chunks > 0: the convolution that produced x1 (single source) left norm1's partial sums in the GroupNorm workspace;
        out_stats: the NEXT op is a GroupNorm over `out` alone (a transformer's norm, or the next resnet's norm1), so conv2
        leaves that norm's partial sums - returns their chunk count (0: not emitted)."""
        cfg, w = self.cfg, self.pk.w
        H, W = self.hw[r.level]
        HW, M = H * W, self.B * H * W
        p = r.prefix
        gn = self._buf("gn", r.level, M, max(r.cin, r.cout))
        t1 = self._buf("res_t1", r.level, M, r.cout)
        assert in_chunks == 0 or x2 is None
        self._gn(ops, x1, x2, gn, w[p + ".norm1.weight"], w[p + ".norm1.bias"], HW, c1, c2, cfg.norm_eps, True, pre_chunks=in_chunks)
        rowbias = self.temb_all.data_ptr() + 2 * self.pk.temb_off[p]
        # conv1 -> norm2: the statistics of t1 come out of conv1's epilogue where a halo-patch kernel runs it (every resnet of the
        # UNet at every size up to 128 pixel tiles per image; DIFFSENSEI_GN_FUSION=0 keeps the three-launch GroupNorm: A/B)
        nch = int(_lib.load().ds_conv3x3_gn_chunks(self.B, H, W, r.cin, r.cout)) if gn_fusion_enabled() else 0
        self.gn_fused = getattr(self, "gn_fused", 0) + (nch > 0)
        self._conv(ops, gn, p + ".conv1", t1, H, W, r.cin, r.cout, rowbias=rowbias, gn_stats=nch > 0)
        self._gn(ops, t1, None, gn, w[p + ".norm2.weight"], w[p + ".norm2.bias"], HW, r.cout, 0, cfg.norm_eps, True, pre_chunks=nch)
        if r.has_shortcut:
            sc = self._buf("res_sc", r.level, M, r.cout)
            self._gemm(ops, x1, w[p + ".conv_shortcut.weight"], sc, M, r.cout, r.cin, bias=w[p + ".conv_shortcut.bias"],
                       x2=x2, K1=c1)
            res = sc
        else:
            assert x2 is None
            res = x1
        nch2 = int(_lib.load().ds_conv3x3_gn_chunks(self.B, H, W, r.cout, r.cout)) if (out_stats and gn_fusion_enabled()) else 0
        self.gn_fused = getattr(self, "gn_fused", 0) + (nch2 > 0)
        self._conv(ops, gn, p + ".conv2", out, H, W, r.cout, r.cout, residual=res, gn_stats=nch2 > 0)
        return nch2

    def _transformer(self, ops, a: AttnSpec, x: Tensor, out: Tensor, in_chunks: int = 0):
        cfg, w, B = self.cfg, self.pk.w, self.B
        H, W = self.hw[a.level]
        N, Cc = H * W, a.channels
        M = B * N
        p = a.prefix
        # V^T rows are read 8 keys at a time: their stride is N rounded up to 8.  The projection then also covers the
        # (< 8) rows after an image's tokens - the next image's first rows, or the zeroed slack behind the last image -
        # and the attention kernel masks those keys; nothing is copied or padded per call.
        Np = (N + 7) // 8 * 8
        tn = self._buf("t_norm", a.level, M + 8, Cc, zero=True)
        # h gets the same 8 rows of zeroed slack as tn: with fused norm1 the transposed to_v GEMM reads the RAW stream h at
        # Np = ceil8(N) rows per image, i.e. up to 7 rows behind the last image (with N % 8 != 0 these became the
        # V^T pad columns, and 0 x NaN of stale bytes would poison the P V MFMA).  Producers only ever write rows < M.
        h = self._buf("t_hidden", a.level, M + 8, Cc, zero=True)
        qk = self._buf("t_qk", a.level, M, 2 * Cc)
        vt = self._buf("t_vt", a.level, B * Np, Cc)
        ao = self._buf("t_attn", a.level, M, Cc)
        q2 = self._buf("t_q2", a.level, M, Cc)
        ff = self._buf("t_ff", a.level, M, 4 * Cc)
        mh, mw = mask_grid_size(N, self.aspect_ratio)
        kvt = self.pk.kv_total
        scale = 1.0 / (Cc // a.heads) ** 0.5
        # LayerNorm fusion (norm1 -> q|k and the transposed to_v, norm2 -> attn2.to_q, norm3 -> GEGLU projection): when every
        # GEMM that writes h and every consumer runs gemm_pp_kernel's branch-free epilogues at this batch, the three LayerNorm
        # launches of every block are replaced by row statistics out of the producer's epilogue + a finalize launch (a few
        # us instead of 76 at batch 64); smaller batches and the 640-channel level keep the LayerNorm kernel
        # Two implementations share one statistics format (ds_gemm_ln_fusable: 1 = gemm_pp_kernel, whose consumers read the
        # (mean, rstd) of a finalize launch; 2 = the 128-wide kernels of small batches and of the 640-channel level, whose
        # consumers sum the partials of their own rows in the epilogue - no launch at all), so every producer / consumer mix works.
        lib = _lib.load()
        kind = lambda m, n, k, epi=0, batch=1: int(lib.ds_gemm_ln_fusable(m, n, k, epi, batch))
        can = lambda m, n, k, epi=0, batch=1: kind(m, n, k, epi, batch) == 1
        have = f"{p}.transformer_blocks.0.attn2.to_q.weight_ln" in w and ln_fusion_enabled()
        # norm2 / norm3: producers = the two out-projections (N = K = C), consumers = attn2.to_q and the GEGLU projection
        # (the GEGLU projection of a small batch may run gemm_g320_kernel: 256 x 320 tiles, one block per CU, its own packing)
        g320 = bool(int(lib.ds_gemm_g320_fits(M, 8 * Cc, Cc, 1)))
        k_proj, k_ff = kind(M, Cc, Cc), kind(M, 8 * Cc, Cc, 4 if g320 else 1)
        pp_min, wide_max, all_pp_min = ln_fusion_limits()
        pays = lambda k: (k == 1 and M * Cc >= pp_min) or (k == 2 and M * Cc <= wide_max)
        fuse = have and pays(k_proj) and pays(k_ff) and not (k_proj == 1 and k_ff == 1 and M * Cc < all_pp_min)
        # norm1 as well: producers = proj_in and the FF down-projection, consumers = q|k and the transposed to_v (operand-swapped
        # form: gemm_pp_kernel needs whole tiles of tokens, the 128-wide kernels take any count)
        k_qk, k_ffd, k_v = kind(M, 2 * Cc, Cc), kind(M, Cc, 4 * Cc), kind(Cc, Np, Cc, 0, B)
        fuse1 = (fuse and f"{p}.transformer_blocks.0.attn1.qk.weight_ln" in w and pays(k_qk) and pays(k_ffd) and pays(k_v)
                 and (k_v == 2 or (Np == N and N % 256 == 0 and can(Cc, N, Cc, 0, B))))   # (operand-swapped: whole tile columns)
        # Statistics entries per row: one per 64 columns, or three per 160 columns where the producers of this level (N = Cc,
        # K = Cc | 4 Cc: proj_in, both out-projections, the FF down-projection) run gemm_t160_kernel - its 160-column tiles hold no
        # whole 64-column strips (csrc/gemm_t160.hip; the rule does not depend on K, so every producer of a level emits one format)
        sw = 160 if (int(lib.ds_gemm_t160_fits(M, Cc, Cc, 1)) and int(lib.ds_gemm_t160_fits(M, Cc, 4 * Cc, 1))) else 64
        nstr = 3 * (Cc // 160) if sw == 160 else Cc // 64
        if fuse:
            part = self._buf32("ln_part", a.level, nstr * M * 2)
            st = self._buf32("ln_stats", a.level, M * 2)
        self.ln_fused_blocks = getattr(self, "ln_fused_blocks", 0) + (a.depth if fuse else 0)
        self.ln_fused_launches = getattr(self, "ln_fused_launches", 0) + a.depth * ((3 if fuse1 else 2) if fuse else 0)
        fin1 = fuse1 and (k_qk == 1 or k_v == 1)     # a finalize launch only where a gemm_pp_kernel consumer reads (mean, rstd)
        self.ln_finalize_launches = getattr(self, "ln_finalize_launches", 0) + a.depth * (
            ((1 if fin1 else 0) + (k_proj == 1) + (k_ff == 1)) if fuse else 0)
        self._gn(ops, x, None, tn, w[p + ".norm.weight"], w[p + ".norm.bias"], N, Cc, 0, 1e-6, False, pre_chunks=in_chunks)
        self._gemm(ops, tn, w[p + ".proj_in.weight"], h, M, Cc, Cc, bias=w[p + ".proj_in.bias"],
                   stats_out=part if fuse1 else None, stats_strip=sw)
        for k in range(a.depth):
            t = f"{p}.transformer_blocks.{k}"
            # ---- attn1 (AttnProcessor2_0): q|k projection, V^T projection, flash attention, out-proj + residual
            if fuse1:    # norm1: statistics of h came out of proj_in / the previous block's FF down-projection
                if fin1:
                    ops.append(make_op("LN_FINALIZE", i=(M, nstr, Cc), f=(1e-5,), p=(part, st)))
                self._gemm(ops, h, w[t + ".attn1.qk.weight_ln"], qk, M, 2 * Cc, Cc, bias=w[t + ".attn1.qk.bias_ln"],
                           ln_stats=st if k_qk == 1 else part, ln_c=w[t + ".attn1.qk.c_ln"], ln_partial=k_qk == 2, ln_nstrips=nstr)
                ops.append(make_op("GEMM", i=(Cc, Np, Cc, Cc, 0, B, 0, 1, 1, int(k_v == 2), nstr), f=(1e-5,),
                                   l=(Cc, 0, Cc, Np, 0, 0, 0, N * Cc, Cc * Np, 0, N, M),
                                   p=(w[t + ".attn1.to_v.weight_ln"], None, h, vt, None, None, None, st if k_v == 1 else part,
                                      w[t + ".attn1.to_v.cb_ln"], None)))
            else:
                ops.append(make_op("LAYERNORM", i=(M, Cc), f=(1e-5,), p=(h, tn, w[t + ".norm1.weight"], w[t + ".norm1.bias"])))
                self._gemm(ops, tn, w[t + ".attn1.qk.weight"], qk, M, 2 * Cc, Cc)
                ops.append(make_op("GEMM", i=(Cc, Np, Cc, Cc, 0, B, 0, 1), l=(Cc, 0, Cc, Np, 0, 0, 0, N * Cc, Cc * Np, 0),
                                   p=(w[t + ".attn1.to_v.weight"], None, tn, vt)))
            ops.append(make_op("SELF_ATTN", i=(B, a.heads, N, N), f=(scale,),
                               l=(2 * Cc, 2 * Cc, Np, Cc, N * 2 * Cc, N * 2 * Cc, N * Cc),
                               p=(qk, qk.data_ptr() + 2 * Cc, vt, ao)))
            self._gemm(ops, ao, w[t + ".attn1.to_out.0.weight"], h, M, Cc, Cc, bias=w[t + ".attn1.to_out.0.bias"],
                       residual=h, stats_out=part if fuse else None, stats_strip=sw)
            # ---- attn2 (MaskedIPAttnProcessor2_0): q projection, fused text+masked-IP attention, out-proj + residual
            if fuse:
                if k_proj == 1:
                    ops.append(make_op("LN_FINALIZE", i=(M, nstr, Cc), f=(1e-5,), p=(part, st)))
                self._gemm(ops, h, w[t + ".attn2.to_q.weight_ln"], q2, M, Cc, Cc, bias=w[t + ".attn2.to_q.bias_ln"],
                           ln_stats=st if k_proj == 1 else part, ln_c=w[t + ".attn2.to_q.c_ln"], ln_partial=k_proj == 2, ln_nstrips=nstr)
            else:
                ops.append(make_op("LAYERNORM", i=(M, Cc), f=(1e-5,), p=(h, tn, w[t + ".norm2.weight"], w[t + ".norm2.bias"])))
                self._gemm(ops, tn, w[t + ".attn2.to_q.weight"], q2, M, Cc, Cc)
            off = self.pk.kv_off[t]
            ops.append(make_op(
                "IP_ATTN",
                i=(B, a.heads, N, cfg.num_text_tokens, cfg.num_ip_tokens, cfg.num_vision_tokens, cfg.num_vision_tokens,
                   cfg.max_num_ips, mh, mw),
                f=(scale, 1.0), l=(Cc, Cc, kvt, LP * kvt, kvt * LP),
                p=(q2, self.k_txt.data_ptr() + 2 * off, self.v_txt.data_ptr() + 2 * off * LP,
                   self.k_ip.data_ptr() + 2 * off, self.v_ip.data_ptr() + 2 * off * LP, self.bbox, ao, self.ip_scale)))
            self._gemm(ops, ao, w[t + ".attn2.to_out.0.weight"], h, M, Cc, Cc, bias=w[t + ".attn2.to_out.0.bias"],
                       residual=h, stats_out=part if fuse else None, stats_strip=sw)
            # ---- GEGLU feed-forward + residual
            if fuse:
                if k_ff == 1:
                    ops.append(make_op("LN_FINALIZE", i=(M, nstr, Cc), f=(1e-5,), p=(part, st)))
                gw, gb, gc = self.pk.geglu320(t, True) if g320 else (
                    w[t + ".ff.net.0.proj.weight_ln"], w[t + ".ff.net.0.proj.bias_ln"], w[t + ".ff.net.0.proj.c_ln"])
                self._gemm(ops, h, gw, ff, M, 8 * Cc, Cc, bias=gb, geglu=320 if g320 else True,
                           ln_stats=st if k_ff == 1 else part, ln_c=gc, ln_partial=k_ff == 2, ln_nstrips=nstr)
            else:
                ops.append(make_op("LAYERNORM", i=(M, Cc), f=(1e-5,), p=(h, tn, w[t + ".norm3.weight"], w[t + ".norm3.bias"])))
                gw, gb, _ = self.pk.geglu320(t, False) if g320 else (w[t + ".ff.net.0.proj.weight"], w[t + ".ff.net.0.proj.bias"], None)
                self._gemm(ops, tn, gw, ff, M, 8 * Cc, Cc, bias=gb, geglu=320 if g320 else True)
            self._gemm(ops, ff, w[t + ".ff.net.2.weight"], h, M, Cc, 4 * Cc, bias=w[t + ".ff.net.2.bias"], residual=h,
                       stats_out=part if (fuse1 and k + 1 < a.depth) else None, stats_strip=sw)
        self._gemm(ops, h, w[p + ".proj_out.weight"], out, M, Cc, Cc, bias=w[p + ".proj_out.bias"], residual=x)

    def _build_forward(self) -> List[DsOp]:
        cfg, pk, B = self.cfg, self.pk, self.B
        w = pk.w
        topo = pk.topo
        ops: List[DsOp] = []
        c0, temb = cfg.block_out_channels[0], cfg.time_embed_dim
        H, W = self.hw[0]
        # ---- 1. time embedding (reference src/models/unet.py:190-199)
        ops.append(make_op("TIMESTEP_EMBED", i=(B, c0, int(cfg.flip_sin_to_cos)), f=(cfg.freq_shift,),
                           p=(self.table, self.ctr, self.t_sin)))
        ops.append(make_op("SKINNY", i=(B, temb, c0, 0, 1),
                           p=(self.t_sin, w["time_embedding.linear_1.weight"], w["time_embedding.linear_1.bias"], None,
                              self.t_h)))
        ops.append(make_op("SKINNY", i=(B, temb, temb, 0, 0),
                           p=(self.t_h, w["time_embedding.linear_2.weight"], w["time_embedding.linear_2.bias"], self.aug,
                              self.emb)))
        ops.append(make_op("SKINNY", i=(B, pk.temb_total, temb, 1, 0),
                           p=(self.emb, w["temb_all.weight"], w["temb_all.bias"], None, self.temb_all)))
        # ---- 2. conv_in + dialog-box embedding (:206-210)
        skips: List[Tuple[Tensor, int, int]] = []  # (tensor, channels, level)

        def new_act(level, ch, tag):
            h_, w_ = self.hw[level]
            return self._alloc((B * h_ * w_, ch), torch.float16)

        x = new_act(0, c0, "conv_in")
        ops.append(make_op("CONV_IN", i=(B, H, W, cfg.in_channels, c0, cfg.max_num_dialogs),
                           p=(self.x_in, w["conv_in.weight"], w["conv_in.bias"], self.dialog_boxes,
                              w["dialog_bbox_embedding"], x)))
        skips.append((x, c0, 0))
        cur, cur_c = x, c0
        # ---- 3. down (:244-265)
        for blk in topo.down:
            lvl = blk["level"]
            nch = 0   # GroupNorm partial chunks the previous op (a resnet's conv2) left for THIS op's first norm
            for j, r in enumerate(blk["resnets"]):
                out = new_act(lvl, r.cout, r.prefix)
                # conv2's output goes straight into a single-source GroupNorm when a transformer follows, or (attention-free
                # level) the block's next resnet: conv2 then emits that norm's statistics (csrc/conv_halo.hip)
                feeds_gn = bool(blk["attns"]) or j + 1 < len(blk["resnets"])
                nch = self._resnet(ops, r, cur, None, cur_c, 0, out, in_chunks=nch, out_stats=feeds_gn)
                cur, cur_c = out, r.cout
                if blk["attns"]:
                    out2 = new_act(lvl, r.cout, blk["attns"][j].prefix)
                    self._transformer(ops, blk["attns"][j], cur, out2, in_chunks=nch)
                    cur = out2
                    nch = 0
                skips.append((cur, cur_c, lvl))
            if blk["downsample"]:
                h_, w_ = self.hw[lvl]
                out = new_act(lvl + 1, cur_c, blk["downsample"])
                self._conv(ops, cur, blk["downsample"], out, h_, w_, cur_c, cur_c, stride=2)
                cur = out
                skips.append((cur, cur_c, lvl + 1))
        # ---- 4. mid (:279-290)
        mid = topo.mid
        lvl = mid["level"]
        out = new_act(lvl, cur_c, "mid0")
        nch = self._resnet(ops, mid["resnets"][0], cur, None, cur_c, 0, out, out_stats=True)
        out2 = new_act(lvl, cur_c, "mid_attn")
        self._transformer(ops, mid["attns"][0], out, out2, in_chunks=nch)
        out3 = new_act(lvl, cur_c, "mid1")
        self._resnet(ops, mid["resnets"][1], out2, None, cur_c, 0, out3)
        cur = out3
        # ---- 5. up (:304-332): the skip concat is never materialised (dual-source GroupNorm / shortcut GEMM)
        for blk in topo.up:
            lvl = blk["level"]
            for j, r in enumerate(blk["resnets"]):
                sk, sk_c, sk_l = skips.pop()
                assert sk_l == lvl and cur_c + sk_c == r.cin, (r.prefix, cur_c, sk_c, r.cin)
                out = new_act(lvl, r.cout, r.prefix)
                nch = self._resnet(ops, r, cur, sk, cur_c, sk_c, out, out_stats=bool(blk["attns"]))
                cur, cur_c = out, r.cout
                if blk["attns"]:
                    out2 = new_act(lvl, r.cout, blk["attns"][j].prefix)
                    self._transformer(ops, blk["attns"][j], cur, out2, in_chunks=nch)
                    cur = out2
            if blk["upsample"]:
                h_, w_ = self.hw[lvl]
                out = new_act(lvl - 1, cur_c, blk["upsample"])
                # Upsample2D resizes to the size of the skip it will meet (== 2x unless a side was odd on the way down)
                self._conv(ops, cur, blk["upsample"], out, h_, w_, cur_c, cur_c, upsample=1, out_hw=self.hw[lvl - 1])
                cur = out
        assert not skips
        # ---- 6. out (:335-338)
        gn = self._buf("gn", 0, B * H * W, c0)
        self._gn(ops, cur, None, gn, w["conv_norm_out.weight"], w["conv_norm_out.bias"], H * W, c0, 0, cfg.norm_eps, True)
        ops.append(make_op("CONV_OUT", i=(B, H, W, c0, cfg.out_channels),
                           p=(gn, w["conv_out.weight"], w["conv_out.bias"], self.eps)))
        return ops

    # -- sampling: forward + CFG + scheduler step + counter advance as ONE replayable plan
    def build_sampler(self, ns: int, kind: int, do_cfg: bool = True):
        """reference src/pipelines/pipeline_diffsensei.py:310-337, one loop iteration per `step_plan.run()`."""
        if self.B != (2 * ns if do_cfg else ns):
            raise ValueError(f"engine batch {self.B} does not match num_samples {ns} (cfg={do_cfg})")
        key = (ns, kind, do_cfg)
        if getattr(self, "_sampler_key", None) == key:
            return
        HW = self.H * self.W
        self.latents = self._alloc((ns, self.cfg.in_channels, self.H, self.W), torch.float16)
        self.prep_plan = Plan([make_op("PREP_INPUT", i=(ns, HW, int(do_cfg)),
                                       p=(self.latents, self.x_in, self.table, self.ctr))], self.keep)
        step_ops = list(self.forward_ops) + [
            make_op("SAMPLER_STEP", i=(ns, HW, kind, int(do_cfg)),
                    p=(self.eps, self.latents, self.x_in, self.table, self.ctr)),
            make_op("ADVANCE", p=(self.ctr,)),
        ]
        self.step_plan = Plan(step_ops, self.keep)
        self._sampler_key = key

    def load_schedule(self, table_rows: Tensor):
        """table_rows: fp32 [n_steps, 8] (see include/diffsensei_hip.h); resets the device step counter."""
        n = table_rows.shape[0]
        if n > self.table.shape[0]:
            raise ValueError("too many steps for the scalar table")
        self.table[:n].copy_(table_rows.to(self.dev, torch.float32))
        self.ctr.zero_()

    # -- host-side setters (tiny H2D copies; never inside a captured graph)
    def set_request(self, encoder_hidden_states: Tensor, text_embeds: Tensor, time_ids: Tensor, bbox: Tensor,
                    dialog_boxes: Optional[Tensor], ip_scale: float):
        self.enc.copy_(encoder_hidden_states.to(self.dev, torch.float16).reshape(self.enc.shape))
        self.text_embeds.copy_(text_embeds.to(self.dev, torch.float16).reshape(self.text_embeds.shape))
        self.time_ids.copy_(time_ids.to(self.dev, torch.float16).reshape(self.time_ids.shape))
        self.bbox.copy_(bbox.to(self.dev, torch.float32).reshape(self.bbox.shape))
        if dialog_boxes is None:
            self.dialog_boxes.zero_()
        else:
            self.dialog_boxes.copy_(dialog_boxes.to(self.dev, torch.int32).reshape(self.dialog_boxes.shape))
        self.ip_scale.fill_(float(ip_scale))
        self.prepare_plan.run()
