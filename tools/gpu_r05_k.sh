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
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -p no:cacheprovider -k "ip_attention or ring or processors" > "$out/k_pytest_ip.log" 2>&1
echo "pytest ip rc=$?"; tail -2 "$out/k_pytest_ip.log"
{ timeout 400 python tools/ipattn_ring_ab.py; timeout 400 python tools/forward_option_ab.py 64 ip_attn_variant 2; } 2>&1 | grep -v amdgpu.ids | tee "$out/r05_ipattn_ring_asm_dma_ab.txt" | cut -c1-220
