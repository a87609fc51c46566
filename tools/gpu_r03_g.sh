#!/usr/bin/env bash
# Round 3, GPU call G: gemm_pp start-skew experiment; counter passes on conv / sp attention / ip_attn (tools/gpu_pmc_r03.sh).
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 400 python tools/pp_skew_ab.py 2>&1 | tail -9 | tee "$out/r03_pp_skew_ab.txt"
timeout 2400 bash tools/gpu_pmc_r03.sh 2>&1 | tail -150
