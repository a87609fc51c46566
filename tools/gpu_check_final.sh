#!/usr/bin/env bash
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
python - <<'PY' || exit 3
from diffsensei_amd import build
import os
assert open(os.path.join(build.LIBDIR, "build.stamp")).read().strip() == build._digest(), "sources changed after the library was built"
PY
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider -s --durations=6 > "$out/r05_pytest_gpu_final_tree.log" 2>&1
echo "pytest rc=$?"; grep -v "amdgpu.ids" "$out/r05_pytest_gpu_final_tree.log" | tail -9
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$out/r05_smoke.log" 2>&1
echo "smoke rc=$?"; tail -1 "$out/r05_smoke.log"
