#!/usr/bin/env bash
# Whole GPU suite + smoke (what the driver runs at round end).
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^\[transformers\]" > "$out/r03_pytest_gpu.log"
echo "pytest rc=$? $(tail -1 $out/r03_pytest_gpu.log)"
grep -a "FAILED\|ERROR" "$out/r03_pytest_gpu.log" | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "^\[transformers\]" | tail -2
