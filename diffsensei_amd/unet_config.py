"""UNet configuration and parameter inventory for the DiffSensei sampling path.

Mirrors the diffusers `UNet2DConditionModel` config keys that the reference's
`UNetMangaModel` (reference: src/models/unet.py:43-86) relies on, plus the three
keys `set_manga_modules` writes (`max_num_ips`, `max_num_dialogs`,
`num_vision_tokens`).  Parameter names follow the diffusers state-dict layout the
reference loads with `unet.load_state_dict(pytorch_model.bin)`
(reference: scripts/demo/gradio_wo_mllm.py:162-169), including
`...attn2.processor.to_k_ip.weight` and `dialog_bbox_embedding`.

No arithmetic lives here: this is the table of names and shapes both the HIP
engine's weight packer and the test oracle are driven from.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, field, asdict
from typing import Dict, List, Tuple


@dataclass
class UNetMangaConfig:
    # diffusers UNet2DConditionModel keys [3P: SDXL-base unet/config.json]
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280)
    down_block_types: Tuple[str, ...] = ("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D")
    up_block_types: Tuple[str, ...] = ("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D")
    layers_per_block: int = 2
    transformer_layers_per_block: Tuple[int, ...] = (1, 2, 10)
    attention_head_dim: Tuple[int, ...] = (5, 10, 20)  # SDXL: this key holds the HEAD COUNT per block
    cross_attention_dim: int = 2048
    addition_time_embed_dim: int = 256
    projection_class_embeddings_input_dim: int = 2816
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    flip_sin_to_cos: bool = True
    freq_shift: int = 0
    sample_size: int = 128
    # written by set_manga_modules (reference: src/models/unet.py:50-53)
    max_num_ips: int = 4
    max_num_dialogs: int = 8
    num_vision_tokens: int = 16
    # text tokens per prompt (CLIP tokenizer max length) – a pipeline property, kept here for sizing
    num_text_tokens: int = 77

    @property
    def time_embed_dim(self) -> int:
        return self.block_out_channels[0] * 4

    @property
    def num_ip_tokens(self) -> int:
        """dummy tokens + max_num_ips*num_vision_tokens (reference: src/models/unet.py:76-77)."""
        return self.num_vision_tokens * (1 + self.max_num_ips)

    def heads(self, level: int) -> int:
        return self.attention_head_dim[level]

    def head_dim(self, level: int) -> int:
        return self.block_out_channels[level] // self.attention_head_dim[level]

    def to_dict(self) -> dict:
        return asdict(self)

    # attribute-style AND dict-style access, like diffusers' FrozenDict config
    def __getitem__(self, k):
        return getattr(self, k)


def sdxl_config() -> UNetMangaConfig:
    return UNetMangaConfig()


def tiny_config() -> UNetMangaConfig:
    """Same topology as SDXL (3 levels, attention-free first level, deep last level) at toy widths."""
    return UNetMangaConfig(
        block_out_channels=(64, 128, 256),
        transformer_layers_per_block=(1, 1, 2),
        attention_head_dim=(1, 2, 4),
        cross_attention_dim=256,
        addition_time_embed_dim=32,
        projection_class_embeddings_input_dim=128 + 6 * 32,  # pooled 128 + 6 time ids
        sample_size=16,
    )


@dataclass
class ResnetSpec:
    prefix: str
    cin: int
    cout: int
    level: int          # resolution level of its input/output (0 = full latent res)
    has_shortcut: bool


@dataclass
class AttnSpec:
    prefix: str
    channels: int
    depth: int
    heads: int
    level: int


@dataclass
class UNetTopology:
    """Execution-ordered description of the UNet (what `forward` walks; reference: src/models/unet.py:244-338)."""
    down: List[dict] = field(default_factory=list)
    mid: dict = field(default_factory=dict)
    up: List[dict] = field(default_factory=list)


def build_topology(cfg: UNetMangaConfig) -> UNetTopology:
    topo = UNetTopology()
    boc = cfg.block_out_channels
    nlev = len(boc)
    # ---- down
    out_ch = boc[0]
    for i, btype in enumerate(cfg.down_block_types):
        in_ch = out_ch
        out_ch = boc[i]
        is_final = i == nlev - 1
        blk = {"type": btype, "resnets": [], "attns": [], "downsample": None, "level": i}
        for j in range(cfg.layers_per_block):
            cin = in_ch if j == 0 else out_ch
            blk["resnets"].append(ResnetSpec(f"down_blocks.{i}.resnets.{j}", cin, out_ch, i, cin != out_ch))
            if btype == "CrossAttnDownBlock2D":
                blk["attns"].append(AttnSpec(f"down_blocks.{i}.attentions.{j}", out_ch,
                                             cfg.transformer_layers_per_block[i], cfg.attention_head_dim[i], i))
        if not is_final:
            blk["downsample"] = f"down_blocks.{i}.downsamplers.0.conv"
        topo.down.append(blk)
    # ---- mid
    c = boc[-1]
    topo.mid = {
        "resnets": [ResnetSpec("mid_block.resnets.0", c, c, nlev - 1, False),
                    ResnetSpec("mid_block.resnets.1", c, c, nlev - 1, False)],
        "attns": [AttnSpec("mid_block.attentions.0", c, cfg.transformer_layers_per_block[-1],
                           cfg.attention_head_dim[-1], nlev - 1)],
        "level": nlev - 1,
    }
    # ---- up
    rev = list(reversed(boc))
    rev_tl = list(reversed(cfg.transformer_layers_per_block))
    rev_heads = list(reversed(cfg.attention_head_dim))
    out_ch = rev[0]
    for i, btype in enumerate(cfg.up_block_types):
        prev_out = out_ch
        out_ch = rev[i]
        in_ch = rev[min(i + 1, nlev - 1)]
        is_final = i == nlev - 1
        level = nlev - 1 - i
        blk = {"type": btype, "resnets": [], "attns": [], "upsample": None, "level": level}
        nres = cfg.layers_per_block + 1
        for j in range(nres):
            res_skip = in_ch if j == nres - 1 else out_ch
            res_in = prev_out if j == 0 else out_ch
            blk["resnets"].append(ResnetSpec(f"up_blocks.{i}.resnets.{j}", res_in + res_skip, out_ch, level, True))
            if btype == "CrossAttnUpBlock2D":
                blk["attns"].append(AttnSpec(f"up_blocks.{i}.attentions.{j}", out_ch, rev_tl[i], rev_heads[i], level))
        if not is_final:
            blk["upsample"] = f"up_blocks.{i}.upsamplers.0.conv"
        topo.up.append(blk)
    return topo


def _resnet_params(p: "OrderedDict[str, Tuple[int, ...]]", r: ResnetSpec, temb: int):
    p[f"{r.prefix}.norm1.weight"] = (r.cin,)
    p[f"{r.prefix}.norm1.bias"] = (r.cin,)
    p[f"{r.prefix}.conv1.weight"] = (r.cout, r.cin, 3, 3)
    p[f"{r.prefix}.conv1.bias"] = (r.cout,)
    p[f"{r.prefix}.time_emb_proj.weight"] = (r.cout, temb)
    p[f"{r.prefix}.time_emb_proj.bias"] = (r.cout,)
    p[f"{r.prefix}.norm2.weight"] = (r.cout,)
    p[f"{r.prefix}.norm2.bias"] = (r.cout,)
    p[f"{r.prefix}.conv2.weight"] = (r.cout, r.cout, 3, 3)
    p[f"{r.prefix}.conv2.bias"] = (r.cout,)
    if r.has_shortcut:
        p[f"{r.prefix}.conv_shortcut.weight"] = (r.cout, r.cin, 1, 1)
        p[f"{r.prefix}.conv_shortcut.bias"] = (r.cout,)


def _attn_params(p, a: AttnSpec, xdim: int):
    c = a.channels
    p[f"{a.prefix}.norm.weight"] = (c,)
    p[f"{a.prefix}.norm.bias"] = (c,)
    p[f"{a.prefix}.proj_in.weight"] = (c, c)
    p[f"{a.prefix}.proj_in.bias"] = (c,)
    for k in range(a.depth):
        t = f"{a.prefix}.transformer_blocks.{k}"
        for n in ("norm1", "norm2", "norm3"):
            p[f"{t}.{n}.weight"] = (c,)
            p[f"{t}.{n}.bias"] = (c,)
        p[f"{t}.attn1.to_q.weight"] = (c, c)
        p[f"{t}.attn1.to_k.weight"] = (c, c)
        p[f"{t}.attn1.to_v.weight"] = (c, c)
        p[f"{t}.attn1.to_out.0.weight"] = (c, c)
        p[f"{t}.attn1.to_out.0.bias"] = (c,)
        p[f"{t}.attn2.to_q.weight"] = (c, c)
        p[f"{t}.attn2.to_k.weight"] = (c, xdim)
        p[f"{t}.attn2.to_v.weight"] = (c, xdim)
        p[f"{t}.attn2.to_out.0.weight"] = (c, c)
        p[f"{t}.attn2.to_out.0.bias"] = (c,)
        # installed by set_manga_modules (reference: src/models/unet.py:70-82)
        p[f"{t}.attn2.processor.to_k_ip.weight"] = (c, xdim)
        p[f"{t}.attn2.processor.to_v_ip.weight"] = (c, xdim)
        p[f"{t}.ff.net.0.proj.weight"] = (8 * c, c)
        p[f"{t}.ff.net.0.proj.bias"] = (8 * c,)
        p[f"{t}.ff.net.2.weight"] = (c, 4 * c)
        p[f"{t}.ff.net.2.bias"] = (c,)
    p[f"{a.prefix}.proj_out.weight"] = (c, c)
    p[f"{a.prefix}.proj_out.bias"] = (c,)


def param_shapes(cfg: UNetMangaConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """name -> shape for every tensor of the UNetMangaModel state dict."""
    topo = build_topology(cfg)
    temb = cfg.time_embed_dim
    c0 = cfg.block_out_channels[0]
    p: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    p["conv_in.weight"] = (c0, cfg.in_channels, 3, 3)
    p["conv_in.bias"] = (c0,)
    p["time_embedding.linear_1.weight"] = (temb, c0)
    p["time_embedding.linear_1.bias"] = (temb,)
    p["time_embedding.linear_2.weight"] = (temb, temb)
    p["time_embedding.linear_2.bias"] = (temb,)
    p["add_embedding.linear_1.weight"] = (temb, cfg.projection_class_embeddings_input_dim)
    p["add_embedding.linear_1.bias"] = (temb,)
    p["add_embedding.linear_2.weight"] = (temb, temb)
    p["add_embedding.linear_2.bias"] = (temb,)
    for blk in topo.down:
        for j, r in enumerate(blk["resnets"]):
            _resnet_params(p, r, temb)
            if blk["attns"]:
                _attn_params(p, blk["attns"][j], cfg.cross_attention_dim)
        if blk["downsample"]:
            c = blk["resnets"][-1].cout
            p[blk["downsample"] + ".weight"] = (c, c, 3, 3)
            p[blk["downsample"] + ".bias"] = (c,)
    _resnet_params(p, topo.mid["resnets"][0], temb)
    _attn_params(p, topo.mid["attns"][0], cfg.cross_attention_dim)
    _resnet_params(p, topo.mid["resnets"][1], temb)
    for blk in topo.up:
        for j, r in enumerate(blk["resnets"]):
            _resnet_params(p, r, temb)
            if blk["attns"]:
                _attn_params(p, blk["attns"][j], cfg.cross_attention_dim)
        if blk["upsample"]:
            c = blk["resnets"][-1].cout
            p[blk["upsample"] + ".weight"] = (c, c, 3, 3)
            p[blk["upsample"] + ".bias"] = (c,)
    p["conv_norm_out.weight"] = (c0,)
    p["conv_norm_out.bias"] = (c0,)
    p["conv_out.weight"] = (cfg.out_channels, c0, 3, 3)
    p["conv_out.bias"] = (cfg.out_channels,)
    p["dialog_bbox_embedding"] = (c0,)
    return p


def random_state_dict(cfg: UNetMangaConfig, seed: int = 0, device="cpu", dtype=None):
    """Seeded synthetic weights at the configured shapes (no checkpoint is reachable offline).  Scales keep fp16
    activations O(1): fan-in normal for matrices (half gain on the residual-branch outputs), near-identity norms.
    Generated on `device` with that device's generator; weight VALUES never affect throughput."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    sd = {}
    for name, shp in param_shapes(cfg).items():
        if name.endswith("norm.weight") or ".norm1.weight" in name or ".norm2.weight" in name \
                or ".norm3.weight" in name or name == "conv_norm_out.weight":
            t = 1.0 + 0.1 * torch.randn(shp, generator=g, device=device)
        elif name.endswith(".bias"):
            t = 0.05 * torch.randn(shp, generator=g, device=device)
        elif name == "dialog_bbox_embedding":
            t = torch.randn(shp, generator=g, device=device)
        else:
            fan_in = 1
            for s in shp[1:]:
                fan_in *= s
            gain = 0.5 if (name.endswith("to_out.0.weight") or name.endswith("ff.net.2.weight")
                           or name.endswith("proj_out.weight") or name.endswith("conv2.weight")) else 1.0
            t = torch.randn(shp, generator=g, device=device) * (gain / fan_in ** 0.5)
        sd[name] = t if dtype is None else t.to(dtype)
    return sd


def attn_processor_names(cfg: UNetMangaConfig) -> List[str]:
    """Keys of `unet.attn_processors` in diffusers order (reference iterates them at src/models/unet.py:58)."""
    names = []
    topo = build_topology(cfg)

    def add(a: AttnSpec):
        for k in range(a.depth):
            names.append(f"{a.prefix}.transformer_blocks.{k}.attn1.processor")
            names.append(f"{a.prefix}.transformer_blocks.{k}.attn2.processor")

    for blk in topo.down:
        for a in blk["attns"]:
            add(a)
    for blk in topo.up:
        for a in blk["attns"]:
            add(a)
    for a in topo.mid["attns"]:
        add(a)
    return names
