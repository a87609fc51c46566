#!/usr/bin/env bash
# The N = 2 control flow of bench.py on ONE GPU (two ranks pinned to device 0, gloo moving the CUDA tensors through the host):
# weight arena + asynchronous slices + checksum verification + barrier-bracketed timing; plus the new interrupt test.
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_call_parity.py -q -m gpu 2>&1 | tail -2
DS_DIST_BACKEND=gloo DS_FORCE_DEVICE=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
   bench.py --gpus 2 --num-samples 4 --steps 1 --warmup 1 > "$out/r03_bench_2ranks_one_gpu.json" 2> "$out/r03_bench_2ranks_one_gpu.err"
echo "rc=$?"; tail -1 "$out/r03_bench_2ranks_one_gpu.json" | cut -c1-1500
tail -5 "$out/r03_bench_2ranks_one_gpu.err" | cut -c1-300
