#!/usr/bin/env bash
# round 5, final part 1: the whole GPU suite (incl. the 2048^2 oracle forward) and smoke on the final tree
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
python - <<'PY' || exit 3
from diffsensei_amd import build
import os
assert open(os.path.join(build.LIBDIR, "build.stamp")).read().strip() == build._digest(), "sources changed after the library was built"
PY
DS_TEST_2048=1 timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider -s --durations=12 > "$out/r05_pytest_gpu_final.log" 2>&1
echo "pytest rc=$?"; grep -v "amdgpu.ids" "$out/r05_pytest_gpu_final.log" | tail -25
grep -h "2048 x 2048 forward\|1536 x 1536 forward\|SDXL 2048\|SDXL 1536\|batch 64 of distinct" "$out/r05_pytest_gpu_final.log" > "$out/r05_unet_2048_vs_oracle.log"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$out/r05_smoke.log" 2>&1
echo "smoke rc=$?"; tail -3 "$out/r05_smoke.log"
