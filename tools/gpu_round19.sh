#!/usr/bin/env bash
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > "$out/run19.log" 2>&1
timeout 900 python tools/gemm_bench.py --variants 2,0 --reps 10 > "$out/gemm_bench19.log" 2>&1
tail -14 "$out/gemm_bench19.log"
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider > "$out/pytest_gpu19.log" 2>&1
echo "pytest rc=$?" | tee -a "$out/run19.log"
tail -4 "$out/pytest_gpu19.log"
timeout 900 python bench.py --steps 2 --warmup 1 --num-samples 4 --no-cpu-baseline > "$out/bench19_ns4.json" 2> "$out/bench19_ns4.err"
echo "bench rc=$?" | tee -a "$out/run19.log"
cut -c1-200 "$out/bench19_ns4.json"
grep -A4 '"gemm\|"self_attn' "$out/bench19_ns4.err" | head -60
