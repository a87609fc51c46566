#!/usr/bin/env bash
# Round 2 end-of-round evidence run: build, the whole GPU suite, smoke, the default bench (cpu_baseline + parity + roofline),
# the rocprofv3 kernel-trace summary of the same workload (graph replay off), the PMC passes on the dominant GEMM, and the
# BASELINE config lines 2 / 3 / 5.
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > "$out/r02_final_run.log" 2>&1
timeout 900 python -m pytest tests -m gpu -q > "$out/r02_final_pytest_gpu.log" 2>&1
echo "pytest rc=$? $(tail -1 $out/r02_final_pytest_gpu.log)" | tee -a "$out/r02_final_run.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> "$out/r02_final_run.log" 2>&1
echo "smoke rc=$?" | tee -a "$out/r02_final_run.log"
timeout 900 python bench.py > "$out/r02_bench_default_ns16_final.json" 2> "$out/r02_bench_default_ns16_final.err"
echo "bench default rc=$?" | tee -a "$out/r02_final_run.log"
tail -1 "$out/r02_bench_default_ns16_final.json" | cut -c1-300
timeout 300 python bench.py --num-samples 1 --refs 1 --no-dialog --steps 3 --warmup 1 --no-cpu-baseline 2> "$out/r02_bench_c2_final.err" \
    | tail -1 | tee "$out/r02_bench_c2_ns1_1ref_final.json" | cut -c1-200
timeout 400 python bench.py --mllm --num-samples 4 --steps 2 --warmup 1 --no-cpu-baseline 2> "$out/r02_bench_c3_final.err" \
    | tail -1 | tee "$out/r02_bench_c3_mllm_ns4_final.json" | cut -c1-200
timeout 400 python bench.py --size 2048 --refs 4 --num-samples 1 --attn fp8 --steps 2 --warmup 1 --no-cpu-baseline 2> "$out/r02_bench_c5_final.err" \
    | tail -1 | tee "$out/r02_bench_c5_2048_ns1_fp8_final.json" | cut -c1-200
cd /tmp
DIFFSENSEI_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/r02_final_prof" -o bench -- \
   python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-roofline \
   > "$GRAFT_REPO_ROOT/$out/r02_final_prof_bench.json" 2> "$GRAFT_REPO_ROOT/$out/r02_final_prof_bench.err"
echo "rocprof rc=$?" | tee -a "$GRAFT_REPO_ROOT/$out/r02_final_run.log"
cd "$GRAFT_REPO_ROOT"
f=$(find "$out/r02_final_prof" -name "*kernel_stats.csv" | head -1)
[[ -n "$f" ]] && cp "$f" "$out/r02_final_kernel_stats.csv" && head -14 "$f" | cut -c1-170
rm -rf "$out/r02_final_prof"
bash tools/gpu_pmc_pp.sh > /dev/null 2>&1
cp "$out/pmc_pp_summary.txt" "$out/r02_pmc_gemm_pp_summary.txt"
tail -30 "$out/r02_pmc_gemm_pp_summary.txt"
