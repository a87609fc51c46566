#!/usr/bin/env bash
# Round 3 evidence run, part E (final tree): the live roofline figures and the rocprofv3 kernel-trace summary of the same workload
# on ONE box (graph replay off under the profiler so that every launch is traced).
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$out/r03_bench_default_ns32_part_e.json" 2> "$out/r03_bench_default_ns32_part_e.err"
echo "bench rc=$?"; tail -1 "$out/r03_bench_default_ns32_part_e.json" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
cd /tmp
DIFFSENSEI_GRAPH=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/r03_final_prof" -o bench -- \
   python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-roofline \
   > "$GRAFT_REPO_ROOT/$out/r03_final_prof_bench.json" 2> "$GRAFT_REPO_ROOT/$out/r03_final_prof_bench.err"
echo "rocprof rc=$?"
cd "$GRAFT_REPO_ROOT"
f=$(find "$out/r03_final_prof" -name "*kernel_stats.csv" | head -1)
[[ -n "$f" ]] && cp "$f" "$out/r03_final_kernel_stats.csv" && head -14 "$f" | cut -c1-170
rm -rf "$out/r03_final_prof"
