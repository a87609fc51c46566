#!/usr/bin/env python
"""Where does the fused LayerNorm pay?  One process, one set of weights: for every UNet batch, the forward (per-op HIP events,
best of 2) with fusion off, with only the gemm_pp_kernel pairs, with only the 128-wide pairs, with both, and under the rule the
launch-plan builder applies by default (engine.LN_FUSION_ALL_PP_MIN_ELEMS).
    python tools/ln_fusion_sweep.py 2 4 8 16 32 64"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsensei_amd import _lib
from diffsensei_amd.unet import UNetMangaModel
from diffsensei_amd.unet_config import sdxl_config

batches = [int(a) for a in sys.argv[1:]] or [2, 8, 32]
lib = _lib.load()
cfg = sdxl_config()
os.environ["DIFFSENSEI_LN_FUSION"] = "1"
m = UNetMangaModel(cfg, device="cuda").init_random(0)
BIG = str(1 << 62)
# (DIFFSENSEI_LN_FUSION, ..._PP_MIN_ELEMS, ..._WIDE_MAX_ELEMS, ..._ALL_PP_MIN_ELEMS); a level fuses only if every GEMM of it may
MODES = {"off": ("0", BIG, "0", "0"), "pp pairs only": ("1", "0", "0", "0"), "128-wide pairs only": ("1", BIG, BIG, "0"),
         "both": ("1", "0", BIG, "0"), "default rule": ("1", None, None, None)}


def forward_ms(eng, reps=2):
    ops = eng.forward_ops
    st = torch.cuda.current_stream()
    best = None
    for rep in range(reps + 1):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(ops) + 1)]
        evs[0].record()
        for k, op in enumerate(ops):
            assert lib.ds_op_run(C.byref(op), st.cuda_stream) == 0, lib.ds_last_error()
            evs[k + 1].record()
        torch.cuda.synchronize()
        t = sum(evs[k].elapsed_time(evs[k + 1]) for k in range(len(ops)))
        if rep:
            best = t if best is None else min(best, t)
    return best


def inputs(B):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 4, 128, 128, generator=g).half().cuda()
    enc = torch.randn(B, 157, cfg.cross_attention_dim, generator=g).half().cuda()
    te, tid = torch.randn(B, 1280, generator=g).half().cuda(), torch.tensor([[1024, 1024, 0, 0, 1024, 1024]] * B).half().cuda()
    bbox = torch.tensor([[[0.05, 0.10, 0.50, 0.95], [0.50, 0.10, 0.95, 0.95], [0, 0, 0, 0], [0, 0, 0, 0]]] * B)
    return x, enc, dict(cross_attention_kwargs={"bbox": bbox, "aspect_ratio": 1.0}, added_cond_kwargs={"text_embeds": te, "time_ids": tid})


x, enc, kw = inputs(2)
m(x, 801.0, enc, **kw)      # the weights (incl. the fused copies) are packed on the first forward: fusion must be on for it
assert any(k.endswith("weight_ln") for k in next(iter(m._engines.values())).pk.w), "fused copies were not packed"
for B in batches:
    x, enc, kw = inputs(B)
    row = []
    for rnd in range(2):
        for name, (on, pp_min, wide_max, all_pp_min) in MODES.items():
            os.environ["DIFFSENSEI_LN_FUSION"] = on
            for key, val in (("PP_MIN", pp_min), ("WIDE_MAX", wide_max), ("ALL_PP_MIN", all_pp_min)):
                if val is None:
                    os.environ.pop(f"DIFFSENSEI_LN_FUSION_{key}_ELEMS", None)
                else:
                    os.environ[f"DIFFSENSEI_LN_FUSION_{key}_ELEMS"] = val
            m._engines.clear()
            m(x, 801.0, enc, **kw)
            eng = next(iter(m._engines.values()))
            row.append((name, forward_ms(eng), len(eng.forward_ops), getattr(eng, "ln_fused_launches", 0),
                        getattr(eng, "ln_finalize_launches", 0)))
    print(f"UNet batch {B} (1024 x 1024):")
    for name in MODES:
        r = [t for t in row if t[0] == name]
        print(f"    {name:22s} forward {min(t[1] for t in r):8.2f} ms  (rounds {[round(t[1], 2) for t in r]})  {r[0][2]} launches, "
              f"{r[0][3]} LayerNorm launches replaced, {r[0][4]} finalize launches")
    sys.stdout.flush()
