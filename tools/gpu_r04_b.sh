#!/usr/bin/env bash
# Round 4, visit B: fused LayerNorm - kernel tests, forward A/B at the benchmark's batch
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 900 python -m pytest -q -m gpu -p no:cacheprovider -s tests/test_gpu_ln_fusion.py > "$out/r04_pytest_ln.log" 2>&1
echo "pytest ln rc=$?"; grep -v "^$" "$out/r04_pytest_ln.log" | tail -12
timeout 600 python tools/ln_fusion_ab.py 64 > "$out/r04_ln_fusion_ab.txt" 2>&1
echo "ab rc=$?"; tail -24 "$out/r04_ln_fusion_ab.txt"
