#!/usr/bin/env python
"""Self-attention kernels per shape on ONE build of the library (DIFFSENSEI_LIB selects it): us per launch of the software-pipelined
kernel (attn_variant 3), the 64-row flash kernel (2) and the automatic dispatch (0), min over ROUNDS x 10 launches, plus the
error of each vs fp32 softmax on the device.  `--compare a.json b.json` prints two builds side by side."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [(64, 20, 1024), (64, 10, 4096), (32, 20, 1024), (16, 20, 1024), (8, 20, 1024), (4, 20, 1024), (2, 20, 1024),
          (8, 10, 4096), (4, 10, 4096), (2, 10, 4096), (2, 20, 4096), (2, 10, 16384), (8, 20, 2304), (2, 10, 9216)]
if len(sys.argv) > 1 and sys.argv[1] == "--compare":
    runs = [json.load(open(p)) for p in sys.argv[2:]]
    for key in runs[0]["rows"]:
        cells = []
        for r in runs:
            row = r["rows"].get(key)
            if row:
                cells.append(f"{r['tag']}: " + " ".join(f"v{v} {row['us'][v]:8.1f} us {row['tf'][v]:6.0f} TF" for v in sorted(row["us"])))
        print(f"{key:22s} " + " | ".join(cells))
    sys.exit(0)
import torch
from diffsensei_amd import _lib, ops
lib = _lib.load()
ROUNDS = int(os.environ.get("ROUNDS", "4"))
tag = os.environ.get("AB_TAG", "base" if os.environ.get("DIFFSENSEI_LIB") else "new")
g = torch.Generator(device="cuda").manual_seed(0)
R = lambda *s: torch.randn(*s, generator=g, device="cuda").half()
rows = {}
for (B, h, N) in SHAPES:
    C = h * 64
    q, k, vt = R(B, N, C) * 2.0, R(B, N, C), R(B, h, 64, N)
    us, tf, err = {}, {}, {}
    ref = None
    if B * h * N * N <= 2 * 20 * 4096 * 4096:      # fp32 reference on the device for the small cases
        qh = q.float().view(B, N, h, 64).transpose(1, 2)
        kh = k.float().view(B, N, h, 64).transpose(1, 2)
        ref = (torch.softmax(qh @ kh.transpose(-1, -2) * 0.125, -1) @ vt.float().transpose(-1, -2)).transpose(1, 2).reshape(B, N, C)
    for var in (3, 2, 0):
        lib.ds_set_option(b"attn_variant", var)
        best = 1e30
        for rnd in range(ROUNDS):
            o = ops.self_attention(q, k, vt, h)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(10):
                ops.self_attention(q, k, vt, h)
            ev[1].record()
            torch.cuda.synchronize()
            best = min(best, ev[0].elapsed_time(ev[1]) * 100)
        us[var], tf[var] = best, 4.0 * B * h * N * N * 64 / best / 1e6
        if ref is not None:
            err[var] = ((o.float() - ref).norm() / ref.norm()).item()
    lib.ds_set_option(b"attn_variant", 0)
    rows[f"B={B} h={h} N={N}"] = {"us": us, "tf": tf, "err": err}
    print(f"[{tag}] B={B:2d} h={h:2d} N={N:5d} " + " ".join(f"v{v}: {us[v]:8.1f} us {tf[v]:6.0f} TF" for v in (3, 2, 0)) +
          ("  rel-L2 " + " ".join(f"v{v} {err[v]:.2e}" for v in err) if err else ""), flush=True)
    del q, k, vt, ref
    torch.cuda.empty_cache()
if len(sys.argv) > 1:
    json.dump({"tag": tag, "rows": rows}, open(sys.argv[1], "w"))
