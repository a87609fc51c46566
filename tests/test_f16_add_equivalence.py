"""CPU: the arithmetic identity the round-5 epilogues rest on.  The residual add of the GEMM / conv epilogues (reference:
`hidden_states = attn.to_out[0](...) + residual` in fp16, src/models/attention_processor.py:84-93 -> diffusers blocks) was
written as (f16)((float)a + (float)b) - convert, convert, f32 add, convert back: 24 instructions per 8 values - and is now one
v_pk_add_f16 per two values.  The two are the same function: the f32 sum of two f16 values is exact unless their exponents
are >= 13 apart, and then the smaller one is far below the larger one's half-ulp, so rounding the f32 sum to f16 never rounds
twice.  Checked here against the exactly rounded sum (f64 add - exact for f16 operands - then ONE rounding to f16) on every
8th finite f16 value paired with EVERY finite f16 value (5.0e8 pairs, a few seconds); DS_TEST_ALL_F16_PAIRS=1 runs all
4 030 726 144 pairs (56 s on 8 cores; done once in round 5: 0 mismatches, profiles/r05_f16_add_equivalence.txt)."""
import os

import numpy as np


def test_f32_sum_of_two_f16_rounds_like_an_f16_add():
    allh = np.arange(65536, dtype=np.uint16).view(np.float16)
    fin = allh[np.isfinite(allh)]
    assert len(fin) == 63488
    a32, a64 = fin.astype(np.float32), fin.astype(np.float64)
    stride = 1 if os.environ.get("DS_TEST_ALL_F16_PAIRS") == "1" else 8
    rows = np.arange(0, len(fin), stride)
    bad = 0
    with np.errstate(over="ignore"):
        for i in range(0, len(rows), 256):
            r = rows[i:i + 256]
            via_f32 = (a32[r, None] + a32[None, :]).astype(np.float16)
            exact = (a64[r, None] + a64[None, :]).astype(np.float16)
            ne = via_f32.view(np.uint16) != exact.view(np.uint16)
            if ne.any():
                ne &= ~((via_f32 == 0) & (exact == 0))      # +0 / -0 spelled differently is not a different number
                bad += int(ne.sum())
    assert bad == 0, f"{bad} of {len(rows) * len(fin)} pairs round differently"
