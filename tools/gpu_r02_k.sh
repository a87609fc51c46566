#!/usr/bin/env bash
# (the llm_gemv_min_cols knob existed only for this run and was removed afterwards: more columns per wavefront is slower)
# Round 2, GPU call K: MLLM token loop with >= 1 / 2 / 3 / 4 weight columns per wavefront (o / down projections)
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
: > "$out/r02_mllm_min_cols_ab.jsonl"
for c in 1 2 3 4 1 2; do
DS_OPTIONS=llm_gemv_min_cols=$c timeout 300 python tools/mllm_bench.py --graph on --new 128 2>/dev/null | tail -1 | sed "s/^/{\"min_cols\": $c, \"run\": /; s/$/}/" | tee -a "$out/r02_mllm_min_cols_ab.jsonl" | cut -c1-260
done
