#!/usr/bin/env python
"""UNet forward at a given batch with an environment switch of the launch-plan builder off / on (same process, same weights,
interleaved): per-kernel HIP-event table of both plans, the difference of the outputs, launches per forward.
    python tools/forward_env_ab.py 64 DIFFSENSEI_SPLIT_RAGGED_N"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsensei_amd import _lib
from diffsensei_amd.unet import UNetMangaModel
from diffsensei_amd.unet_config import sdxl_config

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ENV = sys.argv[2] if len(sys.argv) > 2 else "DIFFSENSEI_LN_FUSION"
lib = _lib.load()
cfg = sdxl_config()
m = UNetMangaModel(cfg, device="cuda").init_random(0)
g = torch.Generator().manual_seed(1)
x = torch.randn(B, 4, 128, 128, generator=g).half().cuda()
enc = torch.randn(B, 157, cfg.cross_attention_dim, generator=g).half().cuda()
te, tid = torch.randn(B, 1280, generator=g).half().cuda(), torch.tensor([[1024, 1024, 0, 0, 1024, 1024]] * B).half().cuda()
bbox = torch.tensor([[[0.05, 0.10, 0.50, 0.95], [0.50, 0.10, 0.95, 0.95], [0, 0, 0, 0], [0, 0, 0, 0]]] * B)
kw = dict(cross_attention_kwargs={"bbox": bbox, "aspect_ratio": 1.0}, added_cond_kwargs={"text_embeds": te, "time_ids": tid})


def table(eng, reps=3):
    ops = eng.forward_ops
    st = torch.cuda.current_stream()
    n = len(ops)
    acc = [0.0] * n
    for rep in range(reps + 1):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        evs[0].record()
        for k, op in enumerate(ops):
            assert lib.ds_op_run(C.byref(op), st.cuda_stream) == 0, lib.ds_last_error()
            evs[k + 1].record()
        torch.cuda.synchronize()
        if rep:
            for k in range(n):
                acc[k] += evs[k].elapsed_time(evs[k + 1])
    t = {}
    name = C.create_string_buffer(96)
    fl, by = C.c_double(), C.c_double()
    for k, op in enumerate(ops):
        lib.ds_op_describe(C.byref(op), name, 96, C.byref(fl), C.byref(by))
        d = t.setdefault(name.value.decode(), [0, 0.0])
        d[0] += 1
        d[1] += acc[k] / reps
    return t, sum(acc) / reps, n


outs, tabs = {}, {}
os.environ[ENV] = "1"      # the packed weights (incl. the fused copies) are built on the first forward
m(x, 801.0, enc, **kw)
for rnd in range(2):
    for mode in ("0", "1"):
        os.environ[ENV] = mode
        m._engines.clear()
        y = m(x, 801.0, enc, **kw).sample
        eng = next(iter(m._engines.values()))
        outs[mode] = y.float()
        tabs.setdefault(mode, []).append(table(eng))
rel = ((outs["1"] - outs["0"]).norm() / outs["0"].norm()).item()
print(f"UNet batch {B}: {ENV}=1 vs 0 output rel-L2 {rel:.3e}")
for mode in ("0", "1"):
    best = min(tabs[mode], key=lambda t: t[1])
    print(f"{ENV}={mode}: forward {best[1]:.2f} ms (rounds: {[round(t[1], 2) for t in tabs[mode]]}), {best[2]} launches")
    for k, (n, ms) in sorted(best[0].items(), key=lambda kv: -kv[1][1])[:9]:
        print(f"    {k:34s} {n:4d} launches {ms:9.3f} ms")
