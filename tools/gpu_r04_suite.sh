#!/usr/bin/env bash
# Round 4: the whole GPU suite + smoke on the current tree
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=12 > "$out/r04_pytest_gpu.log" 2>&1
echo "pytest rc=$?"; tail -25 "$out/r04_pytest_gpu.log"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$out/r04_smoke.log" 2>&1
echo "smoke rc=$?"; tail -3 "$out/r04_smoke.log"
