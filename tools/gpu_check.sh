#!/usr/bin/env bash
# One GPU-box visit: parity tests, smoke, bench, rocprof kernel trace.  Everything lands in gpurun_out/.
# usage (from the repo root, through gpurun):  bash tools/gpu_check.sh [tests|bench|prof|all]
set -u
what="${1:-all}"
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
echo "== $(date) what=$what" | tee "$out/run.log"
rocm-smi --showproductname 2>/dev/null | head -8 >> "$out/run.log"
python -c "import __graft_entry__ as g; g.build()" >> "$out/run.log" 2>&1

if [[ "$what" == "all" || "$what" == "tests" ]]; then
  timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=10 > "$out/pytest_gpu.log" 2>&1
  echo "pytest rc=$?" | tee -a "$out/run.log"
  tail -40 "$out/pytest_gpu.log"
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1
  echo "smoke rc=$?" | tee -a "$out/run.log"
  tail -3 "$out/smoke.log"
fi

if [[ "$what" == "all" || "$what" == "bench" ]]; then
  timeout 1500 python bench.py --steps 2 --warmup 1 --num-samples 4 > "$out/bench_ns4.json" 2> "$out/bench_ns4.err"
  echo "bench ns4 rc=$?" | tee -a "$out/run.log"
  tail -c 3000 "$out/bench_ns4.json"
  timeout 900 python bench.py --steps 2 --warmup 1 --num-samples 1 --no-cpu-baseline > "$out/bench_ns1.json" 2> "$out/bench_ns1.err"
  echo "bench ns1 rc=$?" | tee -a "$out/run.log"
  tail -c 1500 "$out/bench_ns1.json"
fi

if [[ "$what" == "all" || "$what" == "prof" ]]; then
  cd /tmp
  timeout 1200 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$out/prof" -o bench -- \
      python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 1 --num-samples 4 --no-cpu-baseline --no-roofline \
      > "$GRAFT_REPO_ROOT/$out/prof_bench.json" 2> "$GRAFT_REPO_ROOT/$out/prof_bench.err"
  echo "rocprof rc=$?" | tee -a "$GRAFT_REPO_ROOT/$out/run.log"
  cd "$GRAFT_REPO_ROOT"
  find "$out/prof" -name "*kernel_stats*" -o -name "*stats*.csv" | head
  f=$(find "$out/prof" -name "*kernel_stats.csv" | head -1)
  [[ -n "$f" ]] && head -25 "$f"
  # keep only the summaries (the raw trace can be large)
  find "$out/prof" -name "*kernel_trace.csv" -size +20M -delete
fi
echo "== done $(date)" | tee -a "$out/run.log"
