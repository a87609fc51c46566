#!/usr/bin/env python
"""UNet forward at a given batch with ONE switch that changes the launch PLAN flipped between two values - an environment
variable of the plan builder (UPPERCASE: DIFFSENSEI_GN_FUSION) or a library option that the planner's host queries read
(lowercase: gemm_t160) - same process, same weights, plans rebuilt per mode, rounds interleaved: per-kernel HIP-event table of
both plans, the difference of the outputs, launches per forward.
    python tools/forward_plan_ab.py 2 gemm_t160=1,0            python tools/forward_plan_ab.py 64 DIFFSENSEI_GN_FUSION=0,1 [latent]"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsensei_amd import _lib
from diffsensei_amd.unet import UNetMangaModel
from diffsensei_amd.unet_config import sdxl_config

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
KEY, _, vals = (sys.argv[2] if len(sys.argv) > 2 else "DIFFSENSEI_GN_FUSION=0,1").partition("=")
MODES = vals.split(",") if vals else ["0", "1"]
LAT = int(sys.argv[3]) if len(sys.argv) > 3 else 128
lib = _lib.load()
cfg = sdxl_config()
m = UNetMangaModel(cfg, device="cuda").init_random(0)
g = torch.Generator().manual_seed(1)
x = torch.randn(B, 4, LAT, LAT, generator=g).half().cuda()
enc = torch.randn(B, 157, cfg.cross_attention_dim, generator=g).half().cuda()
te, tid = torch.randn(B, 1280, generator=g).half().cuda(), torch.tensor([[LAT * 8, LAT * 8, 0, 0, LAT * 8, LAT * 8]] * B).half().cuda()
bbox = torch.tensor([[[0.05, 0.10, 0.50, 0.95], [0.50, 0.10, 0.95, 0.95], [0, 0, 0, 0], [0, 0, 0, 0]]] * B)
kw = dict(cross_attention_kwargs={"bbox": bbox, "aspect_ratio": 1.0}, added_cond_kwargs={"text_embeds": te, "time_ids": tid})


def set_mode(v):
    if KEY.isupper():
        os.environ[KEY] = v
    else:
        assert lib.ds_set_option(KEY.encode(), int(v)) == 0, lib.ds_last_error()


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
    t, shapes = {}, {}
    name = C.create_string_buffer(96)
    fl, by = C.c_double(), C.c_double()
    for k, op in enumerate(ops):
        lib.ds_op_describe(C.byref(op), name, 96, C.byref(fl), C.byref(by))
        d = t.setdefault(name.value.decode(), [0, 0.0])
        d[0] += 1
        d[1] += acc[k] / reps
        if op.code == 1:      # GEMM: also per shape (M N K, + = residual, s = statistics out, c = LayerNorm consumer)
            key = f"  {name.value.decode()} M={op.i[0]} N={op.i[1]} K={op.i[2]}" + (" batch=%d" % op.i[5] if op.i[5] > 1 else "") + \
                  (" geglu" if op.i[4] == 1 else "") + (" +res" if op.p[6] else "") + (" stats" if op.p[9] else "") + (" ln" if op.p[7] else "")
            d = shapes.setdefault(key, [0, 0.0])
            d[0] += 1
            d[1] += acc[k] / reps
        if op.code == 2:      # CONV3X3: per shape (B H W Cin Cout, s2 = stride 2, up = fused upsample)
            key = f"  {name.value.decode()} conv B={op.i[0]} {op.i[1]}x{op.i[2]} {op.i[3]}->{op.i[4]}" + (" s2" if op.i[5] == 2 else "") + (" up" if op.i[6] else "")
            d = shapes.setdefault(key, [0, 0.0])
            d[0] += 1
            d[1] += acc[k] / reps
    t["__shapes__"] = shapes
    return t, sum(acc) / reps, n


outs, tabs = {}, {}
m(x, 801.0, enc, **kw)      # the packed weights (incl. the fused copies) are built on the first forward
for rnd in range(2):
    for mode in MODES:
        set_mode(mode)
        m._engines.clear()
        y = m(x, 801.0, enc, **kw).sample
        eng = next(iter(m._engines.values()))
        outs[mode] = y.float()
        tabs.setdefault(mode, []).append(table(eng))
rel = ((outs[MODES[1]] - outs[MODES[0]]).norm() / outs[MODES[0]].norm()).item()
print(f"UNet batch {B}, {LAT * 8} x {LAT * 8}: {KEY}={MODES[1]} vs {MODES[0]} output rel-L2 {rel:.3e}, bit-equal {bool(torch.equal(outs[MODES[0]], outs[MODES[1]]))}")
for mode in MODES:
    best = min(tabs[mode], key=lambda t: t[1])
    print(f"{KEY}={mode}: forward {best[1]:.2f} ms (rounds: {[round(t[1], 2) for t in tabs[mode]]}), {best[2]} launches")
    shapes = best[0].pop("__shapes__")
    for k, (n, ms) in sorted(best[0].items(), key=lambda kv: -kv[1][1])[:10]:
        print(f"    {k:34s} {n:4d} launches {ms:9.3f} ms")
    for k, (n, ms) in sorted(shapes.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get('AB_SHAPES', '12'))]:
        print(f"    {k:78s} x{n:4d} {ms / n * 1e3:8.1f} us  {ms:8.3f} ms")
