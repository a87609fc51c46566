#!/usr/bin/env bash
# Round 2, GPU call M: the ring-buffered small-grid GEMM measured where it matters - inside the num_samples-1 pipeline, where
# every GEMM streams COLD weights from HBM (the microbench of call C re-ran one GEMM on MALL-hot operands).
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "ring or gemm_bias" 2>&1 | tail -2
: > "$out/r02_ring_in_pipeline_ab.txt"
for r in 1 0 1 0; do
DS_OPTIONS=gemm_ring=$r timeout 300 python bench.py --num-samples 1 --refs 1 --no-dialog --steps 4 --warmup 1 --no-cpu-baseline 2> "$out/r02_ring_ab_$r.err" \
   | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('gemm_ring=$r (1 = off): %.4f panels/s, %.1f ms per call, forward event sum %.2f ms' % (d['value'], d['ms_per_step'], d['unet_forward']['unet_forward_ms_event_sum']))" | tee -a "$out/r02_ring_in_pipeline_ab.txt"
done
grep -A4 'gemm_glds_kernel<64,false' "$out/r02_ring_ab_0.err" | head -24
