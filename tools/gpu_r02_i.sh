#!/usr/bin/env bash
# (the ip_attn_occupancy knob used below existed only at commit 'conv_in / conv_out walk pixels grid-stride...' and was removed
# after this run - 3 blocks per CU is 20 % slower)
# Round 2, GPU call I: conv_in / conv_out grid-stride (test + effect), ip_attn 3-blocks-per-CU build A/B, and the FULL CPU
# baseline of BASELINE configs[0] (20 steps + VAE decode on the host cores) with 20-step latent + image parity.
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "conv_in or masked_ip or processors" 2>&1 | tail -4 | tee "$out/r02_pytest_i.log"
for occ in 2 3 2 3; do
DS_OPTIONS=ip_attn_occupancy=$occ timeout 100 python tools/one_ipattn.py 32 20 32 32 20 2>&1 | grep "ip_attn B" | sed "s/^/occupancy $occ: /" | tee -a "$out/r02_ipattn_occupancy.txt"
done
timeout 1500 python bench.py --steps 2 --warmup 1 --cpu-full > "$out/r02_bench_ns16_cpu_full.json" 2> "$out/r02_bench_ns16_cpu_full.err"
echo "bench rc=$?"; tail -1 "$out/r02_bench_ns16_cpu_full.json" | cut -c1-300
grep -A3 '"conv_in_kernel"\|"conv_out_kernel"' "$out/r02_bench_ns16_cpu_full.err" | head -12
