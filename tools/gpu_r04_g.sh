#!/usr/bin/env bash
# Round 4, visit G: conv_halo256 tail tiles (Cout = 320) - conv tests + A/B
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 600 python -m pytest -q -m gpu -p no:cacheprovider -x tests/test_gpu_ops.py tests/test_gpu_vae.py -k "conv" > "$out/r04_pytest_conv.log" 2>&1
echo "pytest rc=$?"; tail -4 "$out/r04_pytest_conv.log"
timeout 300 python tools/conv_tail_ab.py 64 > "$out/r04_conv_tail_ab.txt" 2>&1
echo "ab rc=$?"; cat "$out/r04_conv_tail_ab.txt"
