import os, sys
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/diffsensei_amd") else os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from diffsensei_amd import _lib, ops
lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
for (M, N, K, geglu) in [(32768, 10240, 1280, True), (32768, 1280, 5120, False), (32768, 1280, 1280, False), (32768, 2560, 1280, False), (131072, 5120, 640, True)]:
    x = (torch.randn(M, K, generator=g, device="cuda") * 0.5).half()
    w = (torch.randn(N, K, generator=g, device="cuda") * K ** -0.5).half()
    b = (torch.randn(N, generator=g, device="cuda") * 0.5).half()
    res = None if geglu else (torch.randn(M, N, generator=g, device="cuda") * 0.5).half()
    y = torch.empty(M, N // 2 if geglu else N, dtype=torch.float16, device="cuda")
    lib.ds_set_option(b"gemm_variant", 3)
    t = {0: [], 8: []}
    for rnd in range(4):
        for dbg in (0, 8):
            lib.ds_set_option(b"gemm_debug", dbg)
            ops.gemm(x, w, b, residual=res, geglu=geglu, out=y)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(20):
                ops.gemm(x, w, b, residual=res, geglu=geglu, out=y)
            ev[1].record()
            torch.cuda.synchronize()
            t[dbg].append(ev[0].elapsed_time(ev[1]) * 50)
    print(f"M={M} N={N} K={K} geglu={geglu}  setprio: " + " ".join(f"{v:7.1f}" for v in t[0]) + "   no setprio: " + " ".join(f"{v:7.1f}" for v in t[8]), flush=True)
lib.ds_set_option(b"gemm_debug", 0); lib.ds_set_option(b"gemm_variant", 0)
