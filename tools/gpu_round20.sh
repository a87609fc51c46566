#!/usr/bin/env bash
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > "$out/run20.log" 2>&1
timeout 1200 python bench.py > "$out/bench20_default.json" 2> "$out/bench20_default.err"
echo "bench default rc=$?" | tee -a "$out/run20.log"
cat "$out/bench20_default.json"
timeout 900 python tools/res_sweep.py > "$out/res_sweep.log" 2>&1
echo "sweep rc=$?" | tee -a "$out/run20.log"
grep size "$out/res_sweep.log"
cd /tmp
DIFFSENSEI_GRAPH=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/prof20" -o bench -- \
   python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --num-samples 8 --no-cpu-baseline --no-roofline \
   > "$GRAFT_REPO_ROOT/$out/prof20_bench.json" 2> "$GRAFT_REPO_ROOT/$out/prof20_bench.err"
echo "rocprof rc=$?" | tee -a "$GRAFT_REPO_ROOT/$out/run20.log"
cd "$GRAFT_REPO_ROOT"
f=$(find "$out/prof20" -name "*kernel_stats.csv" | head -1)
[[ -n "$f" ]] && head -12 "$f" | cut -c1-160
find "$out/prof20" -name "*kernel_trace.csv" -delete
