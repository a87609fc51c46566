#!/usr/bin/env bash
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > "$out/run2.log" 2>&1
timeout 900 python tools/gemm_bench.py --variants 1,2,3 --reps 10 > "$out/gemm_bench.log" 2>&1
echo "gemm_bench rc=$?" | tee -a "$out/run2.log"
cat "$out/gemm_bench.log" | tail -20
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider -x > "$out/pytest_gpu2.log" 2>&1
echo "pytest rc=$?" | tee -a "$out/run2.log"
tail -15 "$out/pytest_gpu2.log"
timeout 900 python bench.py --steps 2 --warmup 1 --num-samples 4 --no-cpu-baseline > "$out/bench2_ns4.json" 2> "$out/bench2_ns4.err"
echo "bench rc=$?" | tee -a "$out/run2.log"
cat "$out/bench2_ns4.json" | cut -c1-1200
grep -A4 '"gemm\|"self_attn\|"groupnorm\|"ip_attn\|"layernorm' "$out/bench2_ns4.err" | head -80
cd /tmp
DIFFSENSEI_GRAPH=0 timeout 900 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$out/prof2" -o bench -- \
   python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --num-samples 4 --no-cpu-baseline --no-roofline \
   > "$GRAFT_REPO_ROOT/$out/prof2_bench.json" 2> "$GRAFT_REPO_ROOT/$out/prof2_bench.err"
echo "rocprof rc=$?" | tee -a "$GRAFT_REPO_ROOT/$out/run2.log"
cd "$GRAFT_REPO_ROOT"
f=$(find "$out/prof2" -name "*kernel_stats.csv" | head -1)
[[ -n "$f" ]] && head -30 "$f"
find "$out/prof2" -name "*kernel_trace.csv" -size +10M -delete
find "$out/prof2" -name "*.db" -delete
