"""CPU: why BASELINE configs[4]'s "fp8 MFMA attention" is NOT part of this engine (decided in round 6: the e4m3 kernel of rounds
2-5, csrc/attention_fp8.hip, was deleted; the 2048 x 2048 config runs the fp16 kernels).

A host-side model of that kernel's roundings (e4m3 via torch.float8_e4m3fn, everything else fp32) on white-noise operands.  It
reproduces what the GPU measured in rounds 2-5 (rel-L2 5.4e-2 vs fp32 SDPA, profiles/r05_bench_c5_2048_ns1_fp8_final.json) and splits it: quantising Q and
K alone costs 4.0e-2, quantising P and V alone 3.7e-2, P alone 2.5e-2.  So the "hybrid" variant (fp16 Q K^T, fp8 P V: 0.75 of the
fp16 matrix time instead of 0.5) lands at 3.7e-2, and NO variant that feeds e4m3 probabilities to the matrix pipe reaches 2e-2 on
white noise - the floor is the 3-bit mantissa (2^-4 / sqrt 3 relative rms per quantised operand), not the kernel.  On coherent
values (smooth V) the same roundings average out: 1.4e-3.  The north star asks for the reference's fp16 tolerance (2e-3 per op): unreachable.
"""
import pytest
import torch

e4m3 = getattr(torch, "float8_e4m3fn", None)
pytestmark = pytest.mark.skipif(e4m3 is None, reason="torch without float8_e4m3fn")
LOG2E = 1.4426950408889634


def _q8(x):
    return x.to(e4m3).float()


def _case(N=2048, d=64, seed=0):
    g = torch.Generator().manual_seed(seed)
    q, k, v = (torch.randn(N, d, generator=g).half().float() for _ in range(3))
    s = (q @ k.t()) * 0.125
    return q, k, v, s, torch.softmax(s, -1) @ v


def _rel(a, ref):
    return ((a - ref).norm() / ref.norm()).item()


def test_all_e4m3_model_reproduces_the_gpu_measurement():
    q, k, v, s, ref = _case()
    s8 = _q8(q * (0.125 * LOG2E)) @ _q8(k).t()                      # base-2 logits from quantised, pre-scaled Q and K
    p8 = torch.exp2(s8 - s8.max(-1, keepdim=True).values + 8.0)     # the kernel carries 2^8 p
    out = _q8(p8) @ _q8(v) / p8.sum(-1, keepdim=True)               # the row sum uses the unquantised probabilities
    assert 4.5e-2 < _rel(out, ref) < 6.5e-2                         # GPU: 5.4e-2 (tolerance 7e-2 in the GPU test)


def test_error_split_and_the_floor_of_a_hybrid_variant():
    q, k, v, s, ref = _case()
    p = torch.exp2((s - s.max(-1, keepdim=True).values) * LOG2E + 8.0)
    l = p.sum(-1, keepdim=True)
    hybrid = _rel(_q8(p) @ _q8(v) / l, ref)       # exact logits, e4m3 P and V
    p_only = _rel(_q8(p) @ v / l, ref)
    v_only = _rel(p @ _q8(v) / l, ref)
    assert 3.0e-2 < hybrid < 4.5e-2
    assert 2.0e-2 < p_only < 3.2e-2               # already above 2e-2 with V kept in f16
    assert 2.0e-2 < v_only < 3.4e-2
    s8 = _q8(q * (0.125 * LOG2E)) @ _q8(k).t()
    p8 = torch.softmax(s8 / LOG2E, -1)
    assert 3.2e-2 < _rel(p8 @ v, ref) < 5.0e-2    # Q, K alone: the larger half of the total


def test_coherent_values_average_the_rounding_out():
    q, k, v, s, _ = _case()
    N, d = v.shape
    t = torch.linspace(0, 1, N)[:, None]
    vc = (torch.sin(6.28 * t * torch.arange(1, d + 1)[None] / 8) + 2).half().float()
    ref = torch.softmax(s, -1) @ vc
    p = torch.exp2((s - s.max(-1, keepdim=True).values) * LOG2E + 8.0)
    assert _rel(_q8(p) @ _q8(vc) / p.sum(-1, keepdim=True), ref) < 5e-3
