#!/usr/bin/env bash
# Round 4 evidence, part B: PMC passes on the benchmark's dominant launch - the GEGLU projection of the 32 x 32-token level at UNet
# batch 64, which since round 4 is the fused-LayerNorm consumer instantiation gemm_pp_kernel<half,0,9> - and, beside it, the plain
# instantiation on the same shape.
set -u
out=gpurun_out
SHAPE="65536 10240 1280" EPI=geglu_ln bash tools/gpu_pmc_pp.sh
cp "$out/pmc_pp_summary.txt" "$out/r04_pmc_gemm_pp_summary.txt"
python tools/pmc_pp_json.py "$out/r04_pmc_gemm_pp_summary.txt" 65536 10240 1280 geglu_ln > "$out/r04_pmc_gemm_pp.json"
cat "$out/r04_pmc_gemm_pp.json" | head -30
SHAPE="65536 10240 1280" EPI=geglu bash tools/gpu_pmc_pp.sh
cp "$out/pmc_pp_summary.txt" "$out/r04_pmc_gemm_pp_plain_summary.txt"
