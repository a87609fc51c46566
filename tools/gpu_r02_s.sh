#!/usr/bin/env bash
# Round 2, GPU call S: self_attn_kernel with the V^T pieces stored in P-fragment order (one 16-byte read per PV operand).
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "attn or attention" 2>&1 | tail -3
timeout 300 python tools/vendor_ops_ab.py attn 2> /dev/null | tee "$out/r02_self_attn_vt_order.txt"
timeout 300 python tools/attn_bench.py 2> /dev/null | head -4 | tee -a "$out/r02_self_attn_vt_order.txt"
