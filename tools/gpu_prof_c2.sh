#!/usr/bin/env bash
# rocprofv3 kernel-trace summary of ONE whole call of BASELINE configs[1] (1024^2, 50 steps, 1 ref, batch 1; graph replay off so every launch is traced)
set -u
out="$GRAFT_REPO_ROOT/gpurun_out"; mkdir -p "$out"; export TMPDIR=/tmp; cd /tmp
DIFFSENSEI_GRAPH=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/c2_prof" -o bench -- \
   python "$GRAFT_REPO_ROOT/bench.py" --num-samples 1 --refs 1 --no-dialog --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-parity \
   > "$out/r06_prof_c2_bench.json" 2> "$out/r06_prof_c2_bench.err"
echo "rocprof rc=$?"
f=$(find "$out/c2_prof" -name "*kernel_stats.csv" | head -1)
[[ -n "$f" ]] && cp "$f" "$out/r06_c2_ns1_kernel_stats.csv" && head -16 "$f" | cut -c1-150
find "$out/c2_prof" -name "*kernel_trace.csv" -delete
