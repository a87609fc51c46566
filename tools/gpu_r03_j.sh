#!/usr/bin/env bash
# Round 3, GPU call J: ip_attn_kernel with the group-open bits from six scalars vs the previous build (alternate library).
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py -q -m gpu -x -k "ip_attn or masked_ip or MaskedIP or region or unet_forward_vs_oracle" 2>&1 | tail -3
OLD="$PWD/diffsensei_amd/lib/libdiffsensei_hip_oldip.so"
for rnd in 1 2 3; do
  for which in new old; do
    if [ "$which" = old ]; then export DIFFSENSEI_LIB="$OLD"; else unset DIFFSENSEI_LIB; fi
    a=$(python tools/one_ipattn.py 32 20 32 32 20 2>/dev/null | tail -1)
    b=$(python tools/one_ipattn.py 32 10 64 64 20 2>/dev/null | tail -1)
    echo "$which round $rnd: $a | $b" | tee -a "$out/r03_ipattn_groups_ab.txt"
  done
done
