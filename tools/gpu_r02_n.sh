#!/usr/bin/env bash
# Round 2, GPU call N: (1) ring-buffered GEMM for every 64-row grid (gemm_ring=2) vs <= 512 blocks (0) inside the num_samples-1
# and -4 samplers; (2) this repository's kernels vs the PyTorch-ROCm operators the reference would call, same box, same shapes.
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$1: %.4f panels/s, %.1f ms per call, forward event sum %.2f ms' % (d['value'], d['ms_per_step'], d['unet_forward']['unet_forward_ms_event_sum']))"; }
: > "$out/r02_ring_wide_ab.txt"
for r in 0 2 0 2; do
DS_OPTIONS=gemm_ring=$r timeout 300 python bench.py --num-samples 1 --refs 1 --no-dialog --steps 4 --warmup 1 --no-cpu-baseline 2> "$out/r02_ring_wide_ns1_$r.err" \
   | tail -1 | line "num_samples 1 gemm_ring=$r" | tee -a "$out/r02_ring_wide_ab.txt"
done
for r in 0 2; do
DS_OPTIONS=gemm_ring=$r timeout 300 python bench.py --num-samples 4 --steps 3 --warmup 1 --no-cpu-baseline 2> "$out/r02_ring_wide_ns4_$r.err" \
   | tail -1 | line "num_samples 4 gemm_ring=$r" | tee -a "$out/r02_ring_wide_ab.txt"
done
: > "$out/r02_vendor_ops_ab.txt"
for sec in gemm attn norm conv; do
  echo "== $sec" | tee -a "$out/r02_vendor_ops_ab.txt"
  timeout 300 python tools/vendor_ops_ab.py $sec 2> "$out/r02_vendor_$sec.err" | tee -a "$out/r02_vendor_ops_ab.txt"
  tail -3 "$out/r02_vendor_$sec.err"
done
