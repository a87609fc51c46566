#!/usr/bin/env bash
# Round 4, visit E: GroupNorm geometry A/B + the tests that hold GroupNorm
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 600 python -m pytest -q -m gpu -p no:cacheprovider -x tests/test_gpu_ops.py tests/test_gpu_vae.py -k "groupnorm or decoder_engine or gemm_and_groupnorm" > "$out/r04_pytest_gn.log" 2>&1
echo "pytest rc=$?"; tail -5 "$out/r04_pytest_gn.log"
timeout 300 python tools/gn_geometry_ab.py 64 > "$out/r04_gn_geometry_ab.txt" 2>&1
echo "ab rc=$?"; cat "$out/r04_gn_geometry_ab.txt"
timeout 300 python tools/gn_geometry_ab.py 2 > "$out/r04_gn_geometry_ab_b2.txt" 2>&1
echo "ab rc=$?"; cat "$out/r04_gn_geometry_ab_b2.txt"
