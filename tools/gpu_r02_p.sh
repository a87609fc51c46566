#!/usr/bin/env bash
# Round 2, GPU call P: the pp-vs-128x128 dispatch data again now that gemm_pp_kernel's tile hand-over is cheaper
# (UNet batches 6..16), and the same for the level-1 shapes.
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 400 python tools/pp_dispatch_ab.py 2> "$out/r02_pp_dispatch_ab2.err" | tee "$out/r02_pp_dispatch_ab2.txt"
tail -2 "$out/r02_pp_dispatch_ab2.err"
