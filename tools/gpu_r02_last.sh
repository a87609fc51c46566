#!/usr/bin/env bash
# Round 2, last GPU call: PMC passes on the second / third kernels, the two-rank control-flow check on one device, then the
# whole evidence script on the final tree.
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
bash tools/gpu_pmc_ops.sh > /dev/null 2>&1
grep -c "pass" "$out/r02_pmc_conv_attn_summary.txt"
DS_DIST_BACKEND=gloo DS_FORCE_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
   --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 1 --warmup 1 --num-samples 2 --no-cpu-baseline --no-roofline \
   > "$out/r02_bench_2ranks_one_gpu_final.json" 2> "$out/r02_bench_2ranks_one_gpu_final.err"
echo "2-rank rc=$?"; tail -1 "$out/r02_bench_2ranks_one_gpu_final.json" | cut -c1-400
bash tools/gpu_r02_final.sh
