#!/usr/bin/env bash
# Round 3 evidence run, part A: the default bench line (num_samples 32; cpu_baseline + __call__ parity + roofline), the rocprofv3
# kernel-trace summary of the same workload (graph replay off so that every launch is traced), the counter passes on the dominant
# GEMM at the benched shape.
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 1200 python bench.py > "$out/r03_bench_default_ns32_final.json" 2> "$out/r03_bench_default_ns32_final.err"
echo "bench default rc=$?"; tail -1 "$out/r03_bench_default_ns32_final.json" | cut -c1-260
grep -a "^parity" "$out/r03_bench_default_ns32_final.err" | cut -c1-700
cd /tmp
DIFFSENSEI_GRAPH=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/r03_final_prof" -o bench -- \
   python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-roofline \
   > "$GRAFT_REPO_ROOT/$out/r03_final_prof_bench.json" 2> "$GRAFT_REPO_ROOT/$out/r03_final_prof_bench.err"
echo "rocprof rc=$?"
cd "$GRAFT_REPO_ROOT"
f=$(find "$out/r03_final_prof" -name "*kernel_stats.csv" | head -1)
[[ -n "$f" ]] && cp "$f" "$out/r03_final_kernel_stats.csv" && head -12 "$f" | cut -c1-170
rm -rf "$out/r03_final_prof"
SHAPE="65536 10240 1280" bash tools/gpu_pmc_pp.sh > /dev/null 2>&1
cp "$out/pmc_pp_summary.txt" "$out/r03_pmc_gemm_pp_summary.txt"
python tools/pmc_pp_json.py "$out/r03_pmc_gemm_pp_summary.txt" 65536 10240 1280 > "$out/r03_pmc_gemm_pp.json"
grep "traffic_over\|busy\|l2_hit" "$out/r03_pmc_gemm_pp.json"
