#!/usr/bin/env python
"""Debug driver for the software-pipelined attention kernel: the forced-rescale cases, where outputs go wrong."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from diffsensei_amd import _lib, ops
lib = _lib.load()
g = torch.Generator().manual_seed(3 + 300)
B, heads, N = 1, 2, 1088
C = heads * 64
R = lambda: torch.randn(B, N, C, generator=g).half()
q, k, v = R(), R(), R()
k[0, 300, :64] = q[0, 17, :64] * 6.0
k[0, 700, 64:] = q[0, 700, 64:] * 6.0
hs = lambda t: t.float().view(B, N, heads, 64).transpose(1, 2)
vt = v.view(B, N, heads, 64).permute(0, 2, 3, 1).contiguous()
for name, qq in (("spike", q), ("hot", q * 5.0)):
    ref = F.scaled_dot_product_attention(hs(qq), hs(k), hs(v)).transpose(1, 2).reshape(B, N, C)
    for var in (1, 3):
        lib.ds_set_option(b"attn_variant", var)
        got = ops.self_attention(qq.cuda(), k.cuda(), vt.cuda(), heads).float().cpu()
        lib.ds_set_option(b"attn_variant", 0)
        bad = ~torch.isfinite(got)
        rows = bad.any(-1)[0].nonzero().flatten().tolist()
        err = (got - ref).abs().amax(-1)[0]
        err[~torch.isfinite(err)] = -1
        worst = err.topk(8)
        print(f"{name} var{var}: non-finite elements {int(bad.sum())} in rows {rows[:40]}{'...' if len(rows) > 40 else ''}; "
              f"worst finite rows {worst.indices.tolist()} err {[round(x, 4) for x in worst.values.tolist()]}")
        if rows:
            r = rows[0]
            print("   row", r, "cols non-finite:", bad[0, r].nonzero().flatten().tolist()[:70], "values", got[0, r, :8].tolist())
