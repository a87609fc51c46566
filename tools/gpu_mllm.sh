#!/usr/bin/env bash
# MLLM pre-pass evidence: GPU tests of the decode path, decode-rate bench at 13B dims, rocprofv3 kernel stats (eager).
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_gpu_mllm.py -q -m gpu 2>&1 | tail -15 > "$out/mllm_tests.log"
cat "$out/mllm_tests.log"
timeout 100 python tools/mllm_bench.py 2>/dev/null | tail -1 | tee "$out/mllm_bench.json"
cd /tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/mllm_prof" -o mllm -- \
   python "$GRAFT_REPO_ROOT/tools/mllm_bench.py" --graph off --new 48 > /dev/null 2> "$GRAFT_REPO_ROOT/$out/mllm_prof.err"
cd "$GRAFT_REPO_ROOT"
f=$(find "$out/mllm_prof" -name "*kernel_stats.csv" | head -1)
[[ -n "$f" ]] && cp "$f" "$out/mllm_kernel_stats.csv" && head -12 "$f" | cut -c1-200
