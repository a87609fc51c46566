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
{ timeout 400 python tools/gn_geometry_ab.py 64; timeout 400 python tools/forward_option_ab.py 64 gn_variant=2,0; } 2>&1 | grep -v amdgpu.ids | tee "$out/r05_gn_8_in_flight_ab.txt" | cut -c1-260
