#!/usr/bin/env python
"""One UNet forward at a given batch on ONE build of the kernel library (DIFFSENSEI_LIB selects it): per-kernel HIP-event
table, per-GEMM-shape table (every distinct (kernel, M, N, K) of the plan), SHA-1 of the output.  Run once per library from
a shell loop (interleaved rounds) and compare with `tools/forward_lib_ab.py --compare a.json b.json`:
    DIFFSENSEI_LIB=.../libdiffsensei_hip_base.so python tools/forward_lib_ab.py 64 gpurun_out/ab_base_1.json
    python tools/forward_lib_ab.py 64 gpurun_out/ab_new_1.json
The output hash says whether two builds compute the same bits (a restructured main loop must; a changed epilogue fma need not).
"""
import ctypes as C
import hashlib
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def compare(paths):
    runs = [json.load(open(p)) for p in paths]
    tags = sorted({r["tag"] for r in runs})
    best = {t: min((r for r in runs if r["tag"] == t), key=lambda r: r["forward_ms"]) for t in tags}
    for t in tags:
        rs = [r for r in runs if r["tag"] == t]
        print(f"{t:8s} forward {best[t]['forward_ms']:.2f} ms (rounds {[round(r['forward_ms'], 2) for r in rs]}) "
              f"launches {best[t]['launches']} sha1 {sorted({r['sha1'][:12] for r in rs})}")
    if len(tags) == 2:
        a, b = tags
        print(f"\nper kernel (ms, best round of each): {a} -> {b}")
        for k in sorted(set(best[a]["kernels"]) | set(best[b]["kernels"]), key=lambda k: -best[a]["kernels"].get(k, [0, 0])[1]):
            xa, xb = best[a]["kernels"].get(k, [0, 0.0]), best[b]["kernels"].get(k, [0, 0.0])
            if max(xa[1], xb[1]) >= 0.3:
                print(f"  {k:44s} {xa[0]:4d} {xa[1]:9.3f} -> {xb[0]:4d} {xb[1]:9.3f}  ({(xb[1] / xa[1] - 1) * 100 if xa[1] else 0:+.1f} %)")
        print(f"\nper GEMM shape (us per launch, TF/s): {a} -> {b}")
        for k in sorted(set(best[a]["gemms"]) | set(best[b]["gemms"]), key=lambda k: -best[a]["gemms"].get(k, [0, 0, 0])[1]):
            xa, xb = best[a]["gemms"].get(k, [0, 0.0, 0.0]), best[b]["gemms"].get(k, [0, 0.0, 0.0])
            if max(xa[1], xb[1]) >= 1.0:
                ua, ub = xa[1] / max(xa[0], 1) * 1e3, xb[1] / max(xb[0], 1) * 1e3
                fa, fb = xa[2] / max(xa[1], 1e-9) / 1e9, xb[2] / max(xb[1], 1e-9) / 1e9
                print(f"  {k:64s} x{xa[0]:3d} {ua:8.1f} us {fa:6.0f} TF -> {ub:8.1f} us {fb:6.0f} TF  ({(ub / ua - 1) * 100 if ua else 0:+.1f} %)")


if len(sys.argv) > 1 and sys.argv[1] == "--compare":
    compare(sys.argv[2:])
    sys.exit(0)

import torch
from diffsensei_amd import _lib
from diffsensei_amd.unet import UNetMangaModel
from diffsensei_amd.unet_config import sdxl_config

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
out_path = sys.argv[2] if len(sys.argv) > 2 else None
tag = os.environ.get("AB_TAG", "base" if os.environ.get("DIFFSENSEI_LIB") else "new")
S = int(os.environ.get("AB_SIZE", "128"))
lib = _lib.load()
cfg = sdxl_config()
m = UNetMangaModel(cfg, device="cuda").init_random(0)
g = torch.Generator().manual_seed(1)
x = torch.randn(B, 4, S, S, generator=g).half().cuda()
enc = torch.randn(B, 157, cfg.cross_attention_dim, generator=g).half().cuda()
te, tid = torch.randn(B, 1280, generator=g).half().cuda(), torch.tensor([[S * 8, S * 8, 0, 0, S * 8, S * 8]] * B).half().cuda()
bbox = torch.tensor([[[0.05, 0.10, 0.50, 0.95], [0.50, 0.10, 0.95, 0.95], [0, 0, 0, 0], [0, 0, 0, 0]]] * B)
kw = dict(cross_attention_kwargs={"bbox": bbox, "aspect_ratio": 1.0}, added_cond_kwargs={"text_embeds": te, "time_ids": tid})
y = m(x, 801.0, enc, **kw).sample
torch.cuda.synchronize()
sha = hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()
eng = next(iter(m._engines.values()))
ops = eng.forward_ops
st = torch.cuda.current_stream()
n, reps = len(ops), 3
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
kernels, gemms = {}, {}
name = C.create_string_buffer(96)
fl, by = C.c_double(), C.c_double()
for k, op in enumerate(ops):
    lib.ds_op_describe(C.byref(op), name, 96, C.byref(fl), C.byref(by))
    nm = name.value.decode()
    d = kernels.setdefault(nm, [0, 0.0])
    d[0] += 1
    d[1] += acc[k] / reps
    if op.code == _lib.OP["GEMM"]:
        key = f"{nm} M={op.i[0]} N={op.i[1]} K={op.i[2]}" + (" geglu" if op.i[4] else "") + (f" batch={op.i[5]}" if op.i[5] > 1 else "") + \
              (" +res" if op.p[6] else "") + (" stats" if op.p[9] else "")
        e = gemms.setdefault(key, [0, 0.0, 0.0])
        e[0] += 1
        e[1] += acc[k] / reps
        e[2] += fl.value
res = {"tag": tag, "batch": B, "size": S * 8, "forward_ms": sum(acc) / reps, "launches": n, "sha1": sha, "kernels": kernels, "gemms": gemms,
       "lib": os.environ.get("DIFFSENSEI_LIB", "default")}
print(f"[{tag}] UNet batch {B} {S * 8}^2: forward {res['forward_ms']:.2f} ms, {n} launches, sha1 {sha[:12]}")
if out_path:
    json.dump(res, open(out_path, "w"))
