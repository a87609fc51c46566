#!/usr/bin/env bash
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > "$out/run14.log" 2>&1
for ns in 8 2; do
timeout 900 python bench.py --steps 2 --warmup 1 --num-samples $ns --no-cpu-baseline --no-roofline > "$out/bench14_ns$ns.json" 2> "$out/bench14_ns$ns.err"
echo "bench ns=$ns rc=$?" | tee -a "$out/run14.log"
cut -c1-210 "$out/bench14_ns$ns.json"
done
timeout 900 python bench.py --steps 3 --warmup 1 --num-samples 4 > "$out/bench14_ns4.json" 2> "$out/bench14_ns4.err"
echo "bench ns=4 rc=$?" | tee -a "$out/run14.log"
cat "$out/bench14_ns4.json"
cd /tmp
DIFFSENSEI_GRAPH=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/prof14" -o bench -- \
   python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --num-samples 4 --no-cpu-baseline --no-roofline \
   > "$GRAFT_REPO_ROOT/$out/prof14_bench.json" 2> "$GRAFT_REPO_ROOT/$out/prof14_bench.err"
echo "rocprof rc=$?" | tee -a "$GRAFT_REPO_ROOT/$out/run14.log"
cd "$GRAFT_REPO_ROOT"
find "$out/prof14" -type f | head
f=$(find "$out/prof14" -name "*kernel_stats.csv" | head -1)
[[ -n "$f" ]] && head -25 "$f" | cut -c1-220
find "$out/prof14" -name "*kernel_trace.csv" -delete
