#!/usr/bin/env bash
# Round 3 evidence run, part D (final tree: after the gn_finalize, attention-prologue, vae_conv_out and conv_in / conv_out dot2 changes): build, whole GPU suite, smoke,
# the default bench line and the two small-batch configurations.
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^\[transformers\]" > "$out/r03_final_pytest_gpu.log"
echo "pytest rc=$? $(grep -a 'passed\|failed' $out/r03_final_pytest_gpu.log | tail -1)"
grep -a "FAILED\|ERROR" "$out/r03_final_pytest_gpu.log" | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "^\[transformers\]" | tail -1
timeout 1200 python bench.py > "$out/r03_bench_default_ns32_final.json" 2> "$out/r03_bench_default_ns32_final.err"
echo "bench default rc=$?"; tail -1 "$out/r03_bench_default_ns32_final.json" | cut -c1-200
grep -a "^parity" "$out/r03_bench_default_ns32_final.err" | cut -c1-500
timeout 600 python bench.py --num-samples 1 --refs 1 --no-dialog --no-cpu-baseline > "$out/r03_bench_c2_ns1_1ref_final.json" 2> "$out/r03_bench_c2_ns1_1ref_final.err"
echo "bench C2 rc=$?"; tail -1 "$out/r03_bench_c2_ns1_1ref_final.json" | cut -c1-200
timeout 900 python bench.py --mllm --num-samples 4 --no-cpu-baseline > "$out/r03_bench_c3_mllm_ns4_final.json" 2> "$out/r03_bench_c3_mllm_ns4_final.err"
echo "bench C3 rc=$?"; tail -1 "$out/r03_bench_c3_mllm_ns4_final.json" | cut -c1-200
