#!/usr/bin/env python
"""In-situ A/B of launch-time options inside the UNet forward at the benchmark's batch: same plan, same weights, one process,
interleaved rounds; per-option: forward HIP-event sum and the affected kernel's total.
    python tools/forward_option_ab.py [batch] key=a,b [key=a,b ...]      e.g. ip_attn_variant=1,0 gn_variant=1,0 gemm_debug=2048,0"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsensei_amd import _lib
from diffsensei_amd.unet import UNetMangaModel
from diffsensei_amd.unet_config import sdxl_config

args = sys.argv[1:]
B = int(args.pop(0)) if args and args[0].isdigit() else 64
lib = _lib.load()
cfg = sdxl_config()
m = UNetMangaModel(cfg, device="cuda").init_random(0)
g = torch.Generator().manual_seed(1)
x = torch.randn(B, 4, 128, 128, generator=g).half().cuda()
enc = torch.randn(B, 157, cfg.cross_attention_dim, generator=g).half().cuda()
te, tid = torch.randn(B, 1280, generator=g).half().cuda(), torch.tensor([[1024, 1024, 0, 0, 1024, 1024]] * B).half().cuda()
bbox = torch.tensor([[[0.05, 0.10, 0.50, 0.95], [0.50, 0.10, 0.95, 0.95], [0, 0, 0, 0], [0, 0, 0, 0]]] * B)
kw = dict(cross_attention_kwargs={"bbox": bbox, "aspect_ratio": 1.0}, added_cond_kwargs={"text_embeds": te, "time_ids": tid})
y0 = m(x, 801.0, enc, **kw).sample
eng = next(iter(m._engines.values()))
ops = eng.forward_ops
n = len(ops)
name = C.create_string_buffer(96)
fl, by = C.c_double(), C.c_double()
names = []
for op in ops:
    lib.ds_op_describe(C.byref(op), name, 96, C.byref(fl), C.byref(by))
    names.append(name.value.decode())


def run_once():
    st = torch.cuda.current_stream()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    evs[0].record()
    for k, op in enumerate(ops):
        assert lib.ds_op_run(C.byref(op), st.cuda_stream) == 0, lib.ds_last_error()
        evs[k + 1].record()
    torch.cuda.synchronize()
    t = {}
    for k in range(n):
        t[names[k]] = t.get(names[k], 0.0) + evs[k].elapsed_time(evs[k + 1])
    return sum(t.values()), t


run_once()
for spec in args:
    key, vals = spec.split("=")
    vals = [int(v) for v in vals.split(",")]
    res = {v: [] for v in vals}
    for rnd in range(3):
        for v in vals:
            assert lib.ds_set_option(key.encode(), v) == 0, lib.ds_last_error()
            res[v].append(run_once())
    lib.ds_set_option(key.encode(), 0)
    base = min(res[vals[0]], key=lambda r: r[0])
    print(f"{key}:")
    for v in vals:
        best = min(res[v], key=lambda r: r[0])
        diff = sorted(((k, best[1][k] - base[1].get(k, 0.0)) for k in best[1]), key=lambda kv: -abs(kv[1]))[:2]
        print(f"   {key}={v:5d}: forward {best[0]:8.2f} ms (rounds {[round(r[0], 2) for r in res[v]]}); largest per-kernel changes vs {key}={vals[0]}: "
              + ", ".join(f"{k} {d:+.2f} ms" for k, d in diff), flush=True)
