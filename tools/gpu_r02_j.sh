#!/usr/bin/env bash
# Round 2, GPU call J: the N > 1 control flow of bench.py on ONE GPU (2 ranks pinned to device 0, gloo moves the tensors
# through the host): weight broadcast of every engine via broadcast_pipeline + cross-rank checksum, barrier-bracketed timing,
# MAX over ranks.  RCCL itself is covered by tests/test_gpu_rccl.py; the real N = 2/4/8 run belongs to the driver.
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
DS_DIST_BACKEND=gloo DS_FORCE_DEVICE=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
   --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --num-samples 2 --steps 1 --warmup 1 --no-roofline \
   > "$out/r02_bench_2ranks_one_gpu.json" 2> "$out/r02_bench_2ranks_one_gpu.err"
echo "rc=$?"; tail -1 "$out/r02_bench_2ranks_one_gpu.json" | cut -c1-1200; tail -5 "$out/r02_bench_2ranks_one_gpu.err" | cut -c1-300
