#!/usr/bin/env bash
# Round 2, last GPU call: the whole GPU suite, smoke() and a short default bench on the final tree.
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee "$out/r02_verify_pytest.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee "$out/r02_verify_smoke.log"
timeout 600 python bench.py --steps 2 --warmup 1 2> "$out/r02_verify_bench.err" | tail -1 | tee "$out/r02_verify_bench.json" | cut -c1-200
