#!/usr/bin/env bash
# Round 2, GPU call Q: ip_attn_kernel with the character boxes fetched once per block (no loads in the query-tile loop).
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "ip or region or mask or attn" 2>&1 | tail -3
timeout 300 python tools/ipattn_bench.py 2> "$out/r02_ipattn_boxes_once.err" | tee "$out/r02_ipattn_boxes_once.txt"
tail -2 "$out/r02_ipattn_boxes_once.err"
