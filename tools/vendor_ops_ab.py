#!/usr/bin/env python
"""Same-box A/B: this repository's HIP kernels vs the PyTorch-ROCm operators the reference pipeline would call for the same
op on an MI355X (hipBLASLt / rocBLAS behind F.linear, MIOpen behind F.conv2d, the ROCm flash SDPA, ATen norms).  Shapes are
the ones the default bench runs (UNet batch 32, 1024 x 1024).  Interleaved (ours, torch, ours, torch), HIP events, `reps`
launches per number.  The torch side is only a yardstick here - nothing on the product path calls it.

    python tools/vendor_ops_ab.py gemm | conv | attn | norm          (one section per process: MIOpen's search is slow)
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from diffsensei_amd import _lib, ops
from diffsensei_amd.engine import pack_geglu

lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
R = lambda *s: (torch.randn(*s, generator=g, device="cuda") * 0.5).half()


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps * 1e3


def ab(name, ours, theirs, flop, reps=10, check=None):
    t = [[], []]
    for _ in range(2):
        t[0].append(timed(ours, reps))
        t[1].append(timed(theirs, reps))
    a, b = min(t[0]), min(t[1])
    err = ""
    if check is not None:
        y, r = check()
        err = f" | rel-L2 ours vs torch {((y.float() - r.float()).norm() / r.float().norm()).item():.1e}"
    print(f"{name:34s} ours {a:8.1f} us {flop / a / 1e6:6.0f} TF/s | torch {b:8.1f} us {flop / b / 1e6:6.0f} TF/s | "
          f"torch/ours {b / a:5.2f}x{err}", flush=True)


def sec_gemm():
    for name, M, N, K, mode in [("ff1_L2 geglu", 32768, 10240, 1280, "geglu"), ("qk_L2", 32768, 2560, 1280, None),
                                ("out_L2 +res", 32768, 1280, 1280, "res"), ("ff2_L2 +res", 32768, 1280, 5120, "res"),
                                ("ff1_L1 geglu", 131072, 5120, 640, "geglu"), ("ff2_L1 +res", 131072, 640, 2560, "res"),
                                ("qk_L1", 131072, 1280, 640, None), ("out_L1 +res", 131072, 640, 640, "res")]:
        x, w, b = R(M, K), R(N, K) * (K ** -0.5) * 2, R(N)
        res = R(M, N) if mode == "res" else None
        flop = 2.0 * M * N * K
        if mode == "geglu":
            wp, bp = pack_geglu(w, b)
            y = ops.gemm(x, wp, bp, geglu=True)
            ours = lambda: ops.gemm(x, wp, bp, geglu=True, out=y)

            def theirs():
                h, gt = F.linear(x, w, b).chunk(2, dim=-1)
                return h * F.gelu(gt)
            ab(name + " (linear+gelu*mul)", ours, theirs, flop, check=lambda: (ours(), theirs()))
            ab(name + " (vs linear only)", ours, lambda: F.linear(x, w, b), flop)
        elif mode == "res":
            y = ops.gemm(x, w, b, residual=res)
            ours = lambda: ops.gemm(x, w, b, residual=res, out=y)
            theirs = lambda: F.linear(x, w, b) + res
            ab(name + " (linear+add)", ours, theirs, flop, check=lambda: (ours(), theirs()))
            ab(name + " (vs linear only)", ours, lambda: F.linear(x, w, b), flop)
        else:
            y = ops.gemm(x, w, b)
            ours = lambda: ops.gemm(x, w, b, out=y)
            theirs = lambda: F.linear(x, w, b)
            ab(name, ours, theirs, flop, check=lambda: (ours(), theirs()))
        del x, w, b, res, y
        torch.cuda.empty_cache()


def sec_conv():
    torch.backends.cudnn.benchmark = False
    for name, B, H, C in [("conv 1280ch 32x32", 32, 32, 1280), ("conv 640ch 64x64", 32, 64, 640),
                          ("conv 320ch 128x128", 32, 128, 320)]:
        x = R(B, H, H, C)
        w = R(C, 3, 3, C) * ((9 * C) ** -0.5) * 2
        b = R(C)
        xt = x.permute(0, 3, 1, 2)                                  # NCHW view over NHWC memory = channels_last
        wt = w.permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
        xn, wn = xt.contiguous(), w.permute(0, 3, 1, 2).contiguous()   # the reference's own layout: NCHW
        ours = lambda: ops.conv3x3(x, w, b)
        theirs = lambda: F.conv2d(xt, wt, b, padding=1)
        theirs_nchw = lambda: F.conv2d(xn, wn, b, padding=1)
        ab(name + " (torch channels_last)", ours, theirs, 2.0 * B * H * H * C * C * 9, reps=5,
           check=lambda: (ours(), theirs().permute(0, 2, 3, 1)))
        ab(name + " (torch NCHW)", ours, theirs_nchw, 2.0 * B * H * H * C * C * 9, reps=5)
        del x, w, xt, wt, xn, wn
        torch.cuda.empty_cache()


def sec_attn():
    for name, B, heads, N in [("self-attn N=4096 h=10", 32, 10, 4096), ("self-attn N=1024 h=20", 32, 20, 1024)]:
        C = heads * 64
        q, k, v = R(B, N, C), R(B, N, C), R(B, N, C)
        vt = v.view(B, N, heads, 64).permute(0, 2, 3, 1).contiguous()
        qh, kh, vh = (t.view(B, N, heads, 64).transpose(1, 2) for t in (q, k, v))
        ours = lambda: ops.self_attention(q, k, vt, heads)
        theirs = lambda: F.scaled_dot_product_attention(qh, kh, vh)
        ab(name, ours, theirs, 4.0 * B * heads * N * N * 64,
           check=lambda: (ours(), theirs().transpose(1, 2).reshape(B, N, C)))


def sec_norm():
    for name, B, HW, C in [("groupnorm+silu 1280ch 32x32", 32, 1024, 1280), ("groupnorm+silu 320ch 128x128", 32, 16384, 320)]:
        x, gm, bt = R(B, HW, C), R(C), R(C)
        xt = x.permute(0, 2, 1).contiguous()   # [B, C, HW]: the reference's NCHW layout, converted outside the timing
        ours = lambda: ops.groupnorm(x, gm, bt, 32, 1e-5, True)
        theirs = lambda: F.silu(F.group_norm(xt, 32, gm, bt, 1e-5))
        ab(name + " (GB/s in the TF column)", ours, theirs, 4.0 * B * HW * C * 1e3,
           check=lambda: (ours(), theirs().permute(0, 2, 1)))
    for name, M, C in [("layernorm 1280 x 32768 rows", 32768, 1280), ("layernorm 640 x 131072 rows", 131072, 640)]:
        x, gm, bt = R(M, C), R(C), R(C)
        ours = lambda: ops.layernorm(x, gm, bt)
        theirs = lambda: F.layer_norm(x, (C,), gm, bt)
        ab(name + " (GB/s in the TF column)", ours, theirs, 4.0 * M * C * 1e3, check=lambda: (ours(), theirs()))


if __name__ == "__main__":
    {"gemm": sec_gemm, "conv": sec_conv, "attn": sec_attn, "norm": sec_norm}[sys.argv[1]]()
