#!/usr/bin/env bash
# Round 2, GPU call T (last minutes): halo-patch convs with the patch swizzled by column (conflict-free fragment reads)
# vs the row-index swizzle (gemm_debug 1024), same binary, interleaved.
set -u
export TMPDIR=/tmp
timeout 120 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "conv" 2>&1 | tail -2
for r in 1 2; do
DS_OPTIONS=gemm_debug=1024 timeout 40 python tools/one_op.py conv 20 2>/dev/null | sed 's/^/row swizzle    /' | tee -a gpurun_out/r02_conv_col_swizzle.txt
timeout 40 python tools/one_op.py conv 20 2>/dev/null | sed 's/^/column swizzle /' | tee -a gpurun_out/r02_conv_col_swizzle.txt
done
