#!/usr/bin/env bash
# End-of-round evidence run: build, GPU test suite, smoke, default bench (incl. cpu_baseline + roofline), and the
# rocprofv3 kernel-trace summary of the same workload (graph replay off so every launch is traced).
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > "$out/final_run.log" 2>&1
timeout 1400 python -m pytest tests -m gpu -x -q -p no:cacheprovider > "$out/final_pytest_gpu.log" 2>&1
echo "pytest rc=$? $(tail -1 $out/final_pytest_gpu.log)" | tee -a "$out/final_run.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> "$out/final_run.log" 2>&1
echo "smoke rc=$?" | tee -a "$out/final_run.log"
timeout 1500 python bench.py > "$out/final_bench_default.json" 2> "$out/final_bench_default.err"
echo "bench default rc=$?" | tee -a "$out/final_run.log"
tail -1 "$out/final_bench_default.json" | cut -c1-400
cd /tmp
DIFFSENSEI_GRAPH=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/final_prof" -o bench -- \
   python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-roofline \
   > "$GRAFT_REPO_ROOT/$out/final_prof_bench.json" 2> "$GRAFT_REPO_ROOT/$out/final_prof_bench.err"
echo "rocprof rc=$?" | tee -a "$GRAFT_REPO_ROOT/$out/final_run.log"
cd "$GRAFT_REPO_ROOT"
f=$(find "$out/final_prof" -name "*kernel_stats.csv" | head -1)
[[ -n "$f" ]] && cp "$f" "$out/final_kernel_stats.csv" && head -14 "$f" | cut -c1-170
find "$out/final_prof" -name "*kernel_trace.csv" -delete
