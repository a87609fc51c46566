#!/usr/bin/env bash
# round 5, call B: self_attn_sp_kernel without a running maximum, hand-scheduled step, counted LDS waits (asm LDS-DMA)
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > "$out/b_build.log" 2>&1
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -p no:cacheprovider -k "attention or attn or processors" > "$out/b_pytest_attn.log" 2>&1
echo "pytest attn rc=$?"; tail -15 "$out/b_pytest_attn.log"
timeout 900 python -m pytest tests/test_gpu_large_shapes.py -q -m gpu -p no:cacheprovider -s -k "self_attention" > "$out/b_pytest_large.log" 2>&1
echo "pytest large rc=$?"; tail -25 "$out/b_pytest_large.log" | grep -v "variant [012]"
DIFFSENSEI_LIB=$PWD/diffsensei_amd/lib/libdiffsensei_hip_base.so AB_TAG=base timeout 400 python tools/attn_lib_ab.py "$out/attn_base.json" 2>&1 | grep -v amdgpu.ids
AB_TAG=new timeout 400 python tools/attn_lib_ab.py "$out/attn_new.json" 2>&1 | grep -v amdgpu.ids
python tools/attn_lib_ab.py --compare "$out/attn_base.json" "$out/attn_new.json" > "$out/r05_self_attn_sp_ab.txt"
