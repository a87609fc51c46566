#!/usr/bin/env bash
# Round 4, visit D: ip_attn ring variant - bit-identity tests + A/B timing
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 600 python -m pytest -q -m gpu -p no:cacheprovider -x tests/test_gpu_ops.py -k "masked_ip or processors_vs" > "$out/r04_pytest_ip.log" 2>&1
echo "pytest rc=$?"; tail -15 "$out/r04_pytest_ip.log"
timeout 300 python tools/ipattn_ring_ab.py > "$out/r04_ipattn_ring_ab.txt" 2>&1
echo "ab rc=$?"; cat "$out/r04_ipattn_ring_ab.txt"
