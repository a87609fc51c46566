#!/usr/bin/env bash
# MLLM pre-pass evidence: whole GPU test suite, decode-rate bench at 13B dims, rocprofv3 kernel stats (eager).
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 400 python -m pytest tests -q -m gpu 2>&1 | tail -8 > "$out/pytest_gpu_all.log"
cat "$out/pytest_gpu_all.log"
timeout 100 python tools/mllm_bench.py --graph on 2>/dev/null | tail -1 | tee "$out/mllm_bench.json"
cd /tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/mllm_prof" -o mllm -- \
   python "$GRAFT_REPO_ROOT/tools/mllm_bench.py" --graph off --new 48 > /dev/null 2> "$GRAFT_REPO_ROOT/$out/mllm_prof.err"
cd "$GRAFT_REPO_ROOT"
f=$(find "$out/mllm_prof" -name "*kernel_stats.csv" | head -1)
[[ -n "$f" ]] && cp "$f" "$out/mllm_kernel_stats.csv" && head -12 "$f" | cut -c1-200
